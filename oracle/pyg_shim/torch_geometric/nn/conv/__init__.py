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
            if name in ('size_i', 'size_j'):            # PyG special arguments, not gathered tensors
                margs[name] = None
            elif name[-2:] in ('_i', '_j'):
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


class GATConv(MessagePassing):
    """torch_geometric.nn.GATConv restated from its documentation (current single-`lin` layout):
    x' = lin(x) (no bias); alpha_src = <x', att_src>, alpha_dst = <x', att_dst>; self loops removed and
    re-added; e_ij = leaky_relu(alpha_src[j] + alpha_dst[i], 0.2); softmax over the incoming edges of i
    (max-shifted, denominator + 1e-16); out_i = sum_j alpha_ij x'_j (+ bias); heads concatenated."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout=0.0,
                 add_self_loops=True, bias=True, **kwargs):
        kwargs.setdefault('aggr', 'add')
        super().__init__(node_dim=0, **kwargs)
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.add_self_loops = add_self_loops
        self.lin = torch.nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = torch.nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = torch.nn.Parameter(torch.empty(1, heads, out_channels))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(heads * out_channels if concat else out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        from ..inits import glorot, zeros
        glorot(self.lin.weight)
        glorot(self.att_src)
        glorot(self.att_dst)
        zeros(self.bias)

    def forward(self, x, edge_index):
        from ...utils import add_self_loops, remove_self_loops, softmax
        H, C = self.heads, self.out_channels
        n = x.size(0)
        h = self.lin(x).view(-1, H, C)
        a_src = (h * self.att_src).sum(dim=-1)
        a_dst = (h * self.att_dst).sum(dim=-1)
        if self.add_self_loops:
            edge_index, _ = remove_self_loops(edge_index)
            edge_index, _ = add_self_loops(edge_index, num_nodes=n)
        j, i = edge_index[0], edge_index[1]
        e = torch.nn.functional.leaky_relu(a_src[j] + a_dst[i], self.negative_slope)
        alpha = softmax(e, i, num_nodes=n)
        alpha = torch.nn.functional.dropout(alpha, p=self.dropout, training=self.training)
        msg = alpha.unsqueeze(-1) * h[j]
        out = torch.zeros(n, H, C, dtype=msg.dtype).index_add_(0, i, msg)
        out = out.view(-1, H * C) if self.concat else out.mean(dim=1)
        return out if self.bias is None else out + self.bias


class GCNConv(MessagePassing):
    """torch_geometric.nn.GCNConv restated from its documentation (defaults: improved=False, cached=False,
    add_self_loops=True, normalize=True, bias=True): x' = lin(x) (no bias, glorot), gcn_norm over the target
    column, out_i = sum_j norm_ji x'_j, + bias (zeros)."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, add_self_loops=True,
                 normalize=True, bias=True, **kwargs):
        kwargs.setdefault('aggr', 'add')
        super().__init__(**kwargs)
        from ..dense.linear import Linear
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached, self.add_self_loops, self.normalize = improved, cached, add_self_loops, normalize
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer='glorot')
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        from ..inits import glorot, zeros
        glorot(self.lin.weight)
        zeros(self.bias)

    def forward(self, x, edge_index, edge_weight=None):
        from .gcn_conv import gcn_norm
        if self.normalize:
            edge_index, edge_weight = gcn_norm(edge_index, edge_weight, x.size(self.node_dim), self.improved,
                                               self.add_self_loops, self.flow, x.dtype)
        x = self.lin(x)
        out = self.propagate(edge_index, x=x, edge_weight=edge_weight)
        return out if self.bias is None else out + self.bias

    def message(self, x_j, edge_weight):
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j
