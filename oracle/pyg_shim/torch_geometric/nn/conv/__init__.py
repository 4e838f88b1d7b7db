"""MessagePassing.propagate restated: gather(x, edge_index[j]) -> message -> scatter -> update."""
import inspect
import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr='add', *, flow='source_to_target', node_dim=-2, **kwargs):
        super().__init__()
        self.aggr = aggr
        self.flow = flow
        assert flow in ('source_to_target', 'target_to_source')
        self.node_dim = node_dim
        self._msg_params = [p for p in inspect.signature(self.message).parameters]
        self._upd_params = [p for p in inspect.signature(self.update).parameters][1:]

    def message(self, x_j):
        return x_j

    def update(self, inputs):
        return inputs

    def propagate(self, edge_index, size=None, **kwargs):
        if not isinstance(edge_index, torch.Tensor):
            raise NotImplementedError("SparseTensor propagate is out of scope")
        i, j = (1, 0) if self.flow == 'source_to_target' else (0, 1)
        size = [None, None] if size is None else list(size)
        margs = {}
        for name in self._msg_params:
            if name[-2:] in ('_i', '_j'):
                data = kwargs[name[:-2]]
                dim = j if name[-2:] == '_j' else i
                if isinstance(data, (tuple, list)):
                    other = data[1 - dim]
                    if isinstance(other, torch.Tensor) and size[1 - dim] is None:
                        size[1 - dim] = other.size(self.node_dim)
                    data = data[dim]
                if isinstance(data, torch.Tensor):
                    if size[dim] is None:
                        size[dim] = data.size(self.node_dim)
                    data = data.index_select(self.node_dim, edge_index[dim])
                margs[name] = data
            elif name == 'index':
                margs[name] = edge_index[i]
            elif name == 'ptr':
                margs[name] = None
            else:
                margs[name] = kwargs.get(name)
        if size[0] is None:
            size[0] = size[1]
        if size[1] is None:
            size[1] = size[0]
        if 'size_i' in margs:
            margs['size_i'] = size[i]
        if 'size_j' in margs:
            margs['size_j'] = size[j]
        if 'dim_size' in margs:
            margs['dim_size'] = size[i]
        msg = self.message(**margs)
        from ...utils import scatter
        out = scatter(msg, edge_index[i], dim=self.node_dim, dim_size=size[i], reduce=self.aggr)
        return self.update(out, **{k: kwargs.get(k) for k in self._upd_params})


class GATConv(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("GATConv: out of scope for the shim (SURVEY 8(f) rank 1)")


class GCNConv(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("GCNConv: out of scope for the shim")
