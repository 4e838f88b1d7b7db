"""Shared implementation of MagNetConv / MSConv.

Dataflow of the reference (nn/directed/MagNetConv.py:185-249, nn/general/MSConv.py:182-230), with
S_r, S_i the real / imaginary COO operators of the scaled magnetic Laplacian:
    A = sum_k T_k(S_r^T) X_r W_k ,  B = sum_k T_k(S_i^T) X_i W_k          (T_k Chebyshev)
    out_real = A - B + b ,  out_imag = A + B + b
The reference evaluates each of A and B twice (four propagates per order, two are duplicates) and
materialises a [nnz, F] message tensor per propagate.  Here S_r and S_i share ONE sparsity pattern
(off-diagonals of the symmetrised graph + the diagonal; the reference's two self-loop sets
2/lambda and -1 are folded into one diagonal entry, its explicit imaginary zeros dropped into the
same slot), so one fused dual-value HIP SpMM per Chebyshev order produces both T_k parts in a
single traversal; the recurrence 2 S T_{k-1} - T_{k-2} is fused into the kernel epilogue.
"""
import math
import os
from typing import Optional

import numpy as np
import torch
from torch.nn import Parameter

from .. import _cabi
from .. import memo
from ..memo import TensorMemo
from ..message_passing import MessagePassing
from ..dense import FixedSpmm2, MagneticConvFunction, dense_supported, tall_linear
from ..sparse import Pattern, spmm2
from ..utils._laplacian import (assemble_operator_csr, differentiable_parts, fused_operator_csr, laplacian_parts,
                                laplacian_values)

Tensor = torch.Tensor

# PYGSD_GENERIC_OPERATOR_BUILD=1 keeps every build on the generic pipeline (laplacian.hip): the comparison leg of the
# equivalence tests and of tools/build_probe.py
_FUSED_BUILD = os.environ.get("PYGSD_GENERIC_OPERATOR_BUILD", "0") != "1"


def set_fused_build(on: bool) -> bool:
    """Switch the fused operator build on / off; returns the previous setting."""
    global _FUSED_BUILD
    prev, _FUSED_BUILD = _FUSED_BUILD, bool(on)
    return prev


def glorot(t: Optional[Tensor]):
    if t is not None:
        a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
        t.data.uniform_(-a, a)


def zeros(t: Optional[Tensor]):
    if t is not None:
        t.data.fill_(0)


class MagneticOperator:
    """The scaled operator 2L/lambda_max - I in HBM.

    Compute layout (built by pygsd_maglap_assemble_csr, no sorts): ONE int32 CSR over the symmetric
    pattern (E_s off-diagonals + the N diagonal entries, columns ascending) shared by the forward
    (by-target) and backward (by-source) products, with the real / imaginary values of each orientation.
    The COO view (off-diagonals sorted by (row, col), then the diagonal) is kept for the reference-format
    tuple and for the generic autograd path (trainable q)."""

    def __init__(self, csr, values_fwd, values_bwd, off_index, off_real, off_imag, diag, lambda_max, n):
        self.csr, self.values_fwd, self.values_bwd = csr, values_fwd, values_bwd
        # un-scaled Laplacian entries; the COO views below apply 2 x / lambda_max lazily.  An operator from the fused
        # build (`from_fused`) has no COO intermediates: off_index is None and the views are read off the CSR.
        self._off_index, self._off_real, self._off_imag, self._diag = off_index, off_real, off_imag, diag
        self._lambda_max, self.n = lambda_max, n
        self.nnz = csr.nnz if off_real is None else int(off_real.numel()) + n
        self._scaled_cache = None
        self._ref_format = None
        self._coo = None
        self._pattern = None

    @classmethod
    def from_fused(cls, csr, values_fwd, values_bwd, diag, lambda_max, n):
        """Operator built by utils._laplacian.fused_operator_csr: only the compute layout exists."""
        return cls(csr, values_fwd, values_bwd, None, None, None, diag, lambda_max, n)

    def _off_diagonal(self):
        """(off_index int64 [2, E_s], scaled off_real, off_imag) of an operator that only has its CSR: the slots
        whose column differs from their row, in CSR order = sorted by (row, col) = the reference's coalesce order;
        vb_* holds S[row, col] = (2 x) / lambda_max already."""
        rowptr, col = self.csr.rowptr, self.csr.col.long()
        counts = (rowptr[1:] - rowptr[:-1]).long()
        row = torch.repeat_interleave(torch.arange(self.n, dtype=torch.long, device=col.device), counts,
                                      output_size=col.numel())
        off = row != col
        return torch.stack([row[off], col[off]]), self.values_bwd[0][off], self.values_bwd[1][off]

    def _scaled(self):
        """(off_real, off_imag, diag) * 2 / lambda_max with +inf -> 0 (MagNetConv.py:106-107,115-116)."""
        if self._scaled_cache is None:
            def sc(t):
                # tensor / tensor: a true fp32 division like the reference's (its lambda_max is a tensor,
                # MagNetConv.py:98-107) and like csrc's scale_lam -- torch turns `/ python_scalar` into `* (1 / s)`
                lam = torch.as_tensor(self._lambda_max, dtype=t.dtype, device=t.device)
                v = (2.0 * t) / lam
                return v.masked_fill(v == float("inf"), 0)
            if self._off_index is None:
                self._off_index, off_r, off_i = self._off_diagonal()
                self._scaled_cache = (off_r, off_i, sc(self._diag))
            else:
                self._scaled_cache = (sc(self._off_real), sc(self._off_imag), sc(self._diag))
        return self._scaled_cache

    def coo(self):
        """(edge_index [2, E_s + N], values_real, values_imag) with the two reference self-loop sets folded."""
        if self._coo is None:
            off_r, off_i, diag_s = self._scaled()          # (also materialises _off_index of a fused operator)
            loops = torch.arange(self.n, dtype=torch.long, device=self._off_index.device).unsqueeze(0).repeat(2, 1)
            self._coo = (torch.cat([self._off_index, loops], dim=1), torch.cat([off_r, diag_s - 1.0]),
                         torch.cat([off_i, torch.zeros_like(diag_s)]))
        return self._coo

    @property
    def pattern(self):
        """Generic COO-based pattern (both CSR orientations by stable sort) -- only built on demand."""
        if self._pattern is None:
            self._pattern = Pattern(self.coo()[0], self.n, self.n, "source_to_target", validate=False)   # built here
        return self._pattern

    def reference_format(self):
        """(edge_index_real, edge_index_imag, norm_real, norm_imag) exactly as
        MagNetConv.__norm__ returns them (MagNetConv.py:100-120)."""
        if self._ref_format is None:
            off_r, off_i, diag_s = self._scaled()
            dev = self._off_index.device
            loops = torch.arange(self.n, dtype=torch.long, device=dev).unsqueeze(0).repeat(2, 1)
            ei_imag = torch.cat([self._off_index, loops], dim=1)
            ei_real = torch.cat([ei_imag, loops], dim=1)
            norm_real = torch.cat([off_r, diag_s, off_r.new_full((self.n,), -1.0)])
            norm_imag = torch.cat([off_i, off_i.new_zeros(self.n)])
            self._ref_format = (ei_real, ei_imag, norm_real, norm_imag)
        return self._ref_format


class MagneticChebConv(MessagePassing):
    edge_weight_arg = "norm"
    _fused_message = True
    _signed = False

    def _init_common(self, in_channels, out_channels, K, q, trainable_q, normalization, cached, bias,
                     operator_memo=None):
        assert K > 0
        # operator_memo=False: rebuild the operator on every uncached forward exactly like the reference, even for
        # unmodified graph tensors (memo.py documents the contract; None follows PYGSD_NO_OPERATOR_MEMO)
        self._memo_switch = operator_memo
        assert normalization in [None, 'sym'], 'Invalid normalization'
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.normalization = normalization
        self.cached = cached
        self.trainable_q = trainable_q
        if trainable_q:
            self.q = Parameter(torch.Tensor(1).fill_(q))
        else:
            self.q = q
        self.weight = Parameter(torch.Tensor(K + 1, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        zeros(self.bias)
        self._operator = None
        sw = getattr(self, "_memo_switch", None)
        self._op_memo, self._parts_memo, self._lam_memo = TensorMemo(1, sw), TensorMemo(1, sw), TensorMemo(1, sw)
        self.cached_num_edges = None
        self.cached_q = None

    # reference attribute: the 4-tuple built by __norm__ (None until the first forward)
    @property
    def cached_result(self):
        return None if self._operator is None else self._operator.reference_format()

    @cached_result.setter
    def cached_result(self, value):
        if value is not None:
            raise AttributeError("cached_result can only be reset to None")
        self._operator = None

    # ------------------------------------------------------------------------------------------
    def _laplacian_kwargs(self):
        return dict(signed=self._signed, absolute_degree=getattr(self, "absolute_degree", True))

    def _operator_for(self, edge_index, num_nodes, edge_weight, q, normalization, lambda_max, dtype):
        """`cached=False` (the reference default, MagNetConv.py:157-181) rebuilds the operator every forward.
        The operator is a pure function of (edge_index, edge_weight, N, q, normalization, lambda_max), so the
        last one is kept and reused while those inputs are THE SAME TENSORS, unmodified (memo.TensorMemo: weakly
        held identity + in-place version + storage) -- identical values, no re-sort.  Writes that bypass the
        version counter (`.data`) are not seen: disable with `operator_memo=False` / PYGSD_NO_OPERATOR_MEMO=1.
        A trainable q or an edge_weight with gradient recomputes the values every call (they carry the
        gradient); only the sorted structure is reused (`_parts_memo`)."""
        if (isinstance(q, torch.Tensor) and q.requires_grad) or (edge_weight is not None and edge_weight.requires_grad):
            return self._build_operator(edge_index, num_nodes, edge_weight, q, normalization, lambda_max, dtype)
        lam_t = lambda_max if isinstance(lambda_max, torch.Tensor) else None
        tensors = (edge_index, edge_weight, lam_t)
        key = (num_nodes, float(q), normalization, None if lam_t is not None else float(lambda_max), dtype)
        op = self._op_memo.get(tensors, key)
        if op is None:
            op = self._build_operator(edge_index, num_nodes, edge_weight, q, normalization, lambda_max, dtype)
            self._op_memo.put(tensors, key, op)
        return op

    def _parts_for(self, edge_index, edge_weight, num_nodes, dtype):
        key = (num_nodes, dtype)
        parts = self._parts_memo.get((edge_index, edge_weight), key)
        if parts is None:
            parts = laplacian_parts(edge_index, edge_weight, num_nodes, dtype=dtype, **self._laplacian_kwargs())
            self._parts_memo.put((edge_index, edge_weight), key, parts)
        if edge_weight is not None and edge_weight.requires_grad:
            # the reference's Laplacian is differentiable w.r.t. edge_weight: keep the HIP-sorted pattern, recompute
            # its ingredients with autograd-tracked tensor ops (never memoised: they hold this call's graph)
            return differentiable_parts(parts, edge_index, edge_weight, **self._laplacian_kwargs())
        return parts

    def _build_operator(self, edge_index, num_nodes, edge_weight, q, normalization, lambda_max, dtype):
        fixed = not ((isinstance(q, torch.Tensor) and q.requires_grad)
                     or (edge_weight is not None and edge_weight.requires_grad))
        if fixed and _FUSED_BUILD:
            # the case every uncached forward pays for: one fused pass, edge list -> compute layout (csrc/magop.hip)
            qf = float(q.detach().item()) if isinstance(q, torch.Tensor) else float(q)
            built = fused_operator_csr(edge_index, edge_weight, num_nodes, q=qf, normalization=normalization,
                                       lambda_max=float(lambda_max), site=self, **self._laplacian_kwargs())
            if built is not None:
                csr, vf, vb, deg = built
                diag = torch.ones_like(deg) if normalization is not None else deg
                return MagneticOperator.from_fused(csr, vf, vb, diag, lambda_max, num_nodes)
            # a row with more than 4096 symmetrised entries: the generic pipeline has a path for those
        parts = self._parts_for(edge_index, edge_weight, num_nodes, dtype)
        if (isinstance(q, torch.Tensor) and q.requires_grad) or parts.differentiable:
            # trainable q / edge_weight with gradient: generic differentiable route (edge-value gradients by SDDMM)
            off_r, off_i, diag = laplacian_values(parts, q, normalization)
            return MagneticOperator(None, None, None, parts.index, off_r, off_i, diag, lambda_max, num_nodes)
        off_r, off_i, diag, mir_r, mir_i = laplacian_values(parts, q, normalization, mirror=True)
        csr, vf, vb = assemble_operator_csr(parts, off_r, off_i, mir_r, mir_i, diag, float(lambda_max), -1.0)
        return MagneticOperator(csr, vf, vb, parts.index, off_r, off_i, diag, lambda_max, num_nodes)

    def __norm__(self, edge_index, num_nodes, edge_weight, q, normalization, lambda_max, dtype=None):
        """Reference-format operator (MagNetConv.py:78-120): edge_index_real, edge_index_imag,
        edge_weight_real, edge_weight_imag."""
        if not isinstance(lambda_max, torch.Tensor):
            lambda_max = torch.tensor(lambda_max, dtype=dtype or torch.float32, device=edge_index.device)
        return self._build_operator(edge_index, num_nodes, edge_weight, q, normalization, lambda_max,
                                    dtype).reference_format()

    def _lambda_max_eigsh(self, edge_index, edge_weight, num_nodes):
        """normalization=None: largest-magnitude eigenvalue of the un-normalised Laplacian by
        scipy eigsh on the host, as the reference does (get_magnetic_Laplacian.py:88-92)."""
        import scipy.sparse as sp
        from scipy.sparse.linalg import eigsh
        n = int(edge_index.max()) + 1 if num_nodes is None and edge_index.numel() else (num_nodes or 0)
        parts = laplacian_parts(edge_index, edge_weight, n, dtype=torch.float32, **self._laplacian_kwargs())
        off_r, off_i, diag = laplacian_values(parts, self.q, None)
        val = torch.complex(off_r, off_i).cpu().numpy()
        idx = parts.index.cpu().numpy()
        loops = np.arange(n)
        L = sp.coo_matrix((np.concatenate([val, diag.cpu().numpy().astype(np.complex64)]),
                           (np.concatenate([idx[0], loops]), np.concatenate([idx[1], loops]))), (n, n))
        lam = eigsh(L, k=1, which='LM', return_eigenvectors=False)
        return float(np.asarray(lam).real.item())

    def forward(self, x_real, x_imag, edge_index, edge_weight=None, lambda_max=None):
        _cabi.require_gpu(x_real, x_imag, edge_index, edge_weight)
        if x_real.dtype != torch.float32 or x_imag.dtype != torch.float32:
            raise TypeError(f"{type(self).__name__} computes in float32 on the HIP path; got {x_real.dtype}")
        if x_real.dim() > 2:
            # [..., N, F] (the reference's node_dim = -2 propagates and broadcasting matmuls accept leading batch
            # dimensions): one layer evaluation per sample on the SAME operator -- the first builds (or finds) it,
            # the others hit the layer's operator memo / cache
            if x_real.shape != x_imag.shape:
                raise ValueError("x_real and x_imag must have the same shape")
            lead = x_real.shape[:-2]
            flat_r, flat_i = x_real.reshape((-1,) + x_real.shape[-2:]), x_imag.reshape((-1,) + x_imag.shape[-2:])
            outs = [self.forward(flat_r[b], flat_i[b], edge_index, edge_weight, lambda_max) for b in range(flat_r.size(0))]
            out_r = torch.stack([o[0] for o in outs]).reshape(lead + outs[0][0].shape)
            out_i = torch.stack([o[1] for o in outs]).reshape(lead + outs[0][1].shape)
            return out_r, out_i
        if x_real.dim() != 2 or x_imag.dim() != 2:
            raise ValueError(f"{type(self).__name__}: inputs must be [N, F] or [..., N, F], got {tuple(x_real.shape)}")
        if self.trainable_q:
            self.q = Parameter(torch.clamp(self.q, 0, 0.25))

        if self.cached and self._operator is not None:
            if edge_index.size(1) != self.cached_num_edges:
                raise RuntimeError(
                    'Cached {} number of edges, but found {}. Please '
                    'disable the caching behavior of this layer by removing '
                    'the `cached=True` argument in its constructor.'.format(
                        self.cached_num_edges, edge_index.size(1)))
            if self.q != self.cached_q:
                raise RuntimeError(
                    'Cached q is {}, but found {} in input. Please '
                    'disable the caching behavior of this layer by removing '
                    'the `cached=True` argument in its constructor.'.format(
                        self.cached_q, self.q))
        if not self.cached or self._operator is None:
            self.cached_num_edges = edge_index.size(1)
            if self.trainable_q:
                self.cached_q = self.q.detach().item()
            else:
                self.cached_q = self.q
            if self.normalization != 'sym' and lambda_max is None:
                if self.trainable_q:
                    raise RuntimeError(
                        'Cannot train q while not calculating maximum eigenvalue of Laplacian!')
                lambda_max = self._lam_memo.get((edge_index, edge_weight), float(self.q))
                if lambda_max is None:                           # same graph tensors: same eigenvalue
                    lambda_max = self._lam_memo.put((edge_index, edge_weight), float(self.q),
                                                    self._lambda_max_eigsh(edge_index, edge_weight, None))
            if lambda_max is None:
                lambda_max = 2.0
            with memo.verified(edge_index, edge_weight, lambda_max if isinstance(lambda_max, torch.Tensor) else None):
                self._operator = self._operator_for(edge_index, x_real.size(self.node_dim), edge_weight,
                                                    self.q, self.normalization, lambda_max, x_real.dtype)

        op = self._operator
        fixed = op.csr is not None            # operator values carry no gradient (q not trainable)
        wide = self.weight.size(0) > 1 and self.in_channels >= 2 * self.out_channels
        if (fixed and not wide and x_real.dim() == 2
                and dense_supported(self.in_channels, self.out_channels, self.weight.size(0))):
            # whole layer as one autograd node: K dual SpMMs + one MFMA dense pass each way
            return MagneticConvFunction.apply(x_real, x_imag, self.weight, self.bias, op)
        # general path (trainable q -> edge-value gradients through SDDMM; shapes the MFMA kernels
        # do not tile): same HIP SpMMs, dense stage composed from library GEMMs
        if fixed:
            def prop(xa, xb, za=None, zb=None, alpha=1.0, beta=0.0):
                return FixedSpmm2.apply(xa, xb, za, zb, op, alpha, beta)
        else:
            _, w_r, w_i = op.coo()

            def prop(xa, xb, za=None, zb=None, alpha=1.0, beta=0.0):
                return spmm2(op.pattern, xa, xb, w_r, w_i, za=za, zb=zb, alpha=alpha, beta=beta)
        k1 = self.weight.size(0)
        if k1 > 1 and self.in_channels >= 2 * self.out_channels:
            # Wide inputs (raw first-layer features, e.g. 2879 -> 16): evaluate sum_k T_k(S) (X W_k) by the
            # Clenshaw recurrence b_k = X W_k + 2 S b_{k+1} - b_{k+2}, result = X W_0 + S b_1 - b_2, so the K
            # SpMMs run at width F_out instead of F_in (the reference propagates at F_in, MagNetConv.py:196-199;
            # same polynomial, re-associated -- within the 1e-5 bar, checked on the wide golden fixtures).
            ya = [tall_linear(x_real, self.weight[k]) for k in range(k1)]
            yb = [tall_linear(x_imag, self.weight[k]) for k in range(k1)]
            b1_a, b1_b, b2_a, b2_b = ya[k1 - 1], yb[k1 - 1], None, None
            for k in range(k1 - 2, 0, -1):
                za = ya[k] if b2_a is None else ya[k] - b2_a
                zb = yb[k] if b2_b is None else yb[k] - b2_b
                nb_a, nb_b = prop(b1_a, b1_b, za, zb, 2.0, 1.0)
                b2_a, b2_b, b1_a, b1_b = b1_a, b1_b, nb_a, nb_b
            za = ya[0] if b2_a is None else ya[0] - b2_a
            zb = yb[0] if b2_b is None else yb[0] - b2_b
            acc_a, acc_b = prop(b1_a, b1_b, za, zb, 1.0, 1.0)
            out_real = acc_a - acc_b
            out_imag = acc_a + acc_b
            if self.bias is not None:
                out_real += self.bias
                out_imag += self.bias
            return out_real, out_imag
        # A-chain on (S_r, X_r), B-chain on (S_i, X_i); one fused traversal per Chebyshev order
        t0_r, t0_i = x_real, x_imag
        acc_a = tall_linear(t0_r, self.weight[0])
        acc_b = tall_linear(t0_i, self.weight[0])
        if self.weight.size(0) > 1:
            t1_r, t1_i = prop(t0_r, t0_i)
            acc_a = acc_a + tall_linear(t1_r, self.weight[1])
            acc_b = acc_b + tall_linear(t1_i, self.weight[1])
        for k in range(2, self.weight.size(0)):
            t2_r, t2_i = prop(t1_r, t1_i, t0_r, t0_i, 2.0, -1.0)
            acc_a = acc_a + tall_linear(t2_r, self.weight[k])
            acc_b = acc_b + tall_linear(t2_i, self.weight[k])
            t0_r, t0_i, t1_r, t1_i = t1_r, t1_i, t2_r, t2_i

        out_real = acc_a - acc_b
        out_imag = acc_a + acc_b
        if self.bias is not None:
            out_real += self.bias
            out_imag += self.bias
        return out_real, out_imag

    def message(self, x_j, norm):
        return norm.view(-1, 1) * x_j

    def __repr__(self):
        return '{}({}, {}, filter size={}, normalization={})'.format(
            self.__class__.__name__, self.in_channels, self.out_channels,
            self.weight.size(0), self.normalization)
