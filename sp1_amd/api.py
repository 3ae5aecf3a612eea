"""Thin Python mirror of the reference's operator interfaces for the core-shard commit/open path,
calling the HIP backend through the C ABI (include/sp1hip.h). PyTorch is used only as the owner of
device memory (int32 tensors reinterpreted as u32 words) and for streams; every computation is a
hand-written gfx950 kernel inside libsp1hip.so. There is no CPU fallback.

Names follow the reference:
  DuplexChallenger                 slop_challenger::DuplexChallenger      (/root/reference/slop/crates/challenger/src/lib.rs:L25-L87)
  DftEncoder.encode_batch          CpuDftEncoder::encode_batch            (/root/reference/slop/crates/basefold-prover/src/encoder.rs:L22-L38)
  MerkleTcsProver.commit_tensors / prove_openings_at_indices / compute_openings_at_indices
                                   TensorCsProver / ComputeTcsOpenings    (/root/reference/slop/crates/merkle-tree/src/tcs.rs:L15-L47)
  BasefoldProver.commit_mles / prove_trusted_mle_evaluations
                                   BasefoldProver                         (/root/reference/slop/crates/basefold-prover/src/prover.rs:L78-L243)
Device layouts: base tensors column-major [height x width]; ext vectors SoA (see include/sp1hip.h).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Ext, FriConfig, GkrChip, ShardChip, ShardParams, Table, Tensor, ZcChip, check

P = 0x7F000001


def _L():
    return _lib.load()


def _stream_ptr(stream=None):
    if stream is None:
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def _dptr(t):
    return C.c_void_p(t.data_ptr())


def device_words(n, device=None):
    """Uninitialised device buffer of n u32 words."""
    return torch.empty(int(n), dtype=torch.int32, device=device or torch.device("cuda", torch.cuda.current_device()))


def to_device(a, device=None):
    """numpy uint32 array -> flat device word buffer (same memory order)."""
    a = np.ascontiguousarray(a, dtype=np.uint32)
    t = torch.from_numpy(a.view(np.int32).reshape(-1))
    return t.to(device or torch.device("cuda", torch.cuda.current_device()), non_blocking=False)


def to_host(t, shape=None):
    a = t.detach().cpu().numpy().view(np.uint32)
    return a.reshape(shape) if shape is not None else a


def _ext(x):
    e = Ext()
    for k in range(4):
        e.c[k] = int(x[k])
    return e


def _ext_array(xs):
    xs = np.asarray(xs, dtype=np.uint32).reshape(-1, 4)
    arr = (Ext * max(len(xs), 1))()
    for i, x in enumerate(xs):
        for k in range(4):
            arr[i].c[k] = int(x[k])
    return arr


class ColMajor:
    """A column-major [height x width] base-field tensor in device memory (Montgomery words)."""

    def __init__(self, words, height, width):
        assert words.numel() == height * width and words.dtype == torch.int32
        self.words, self.height, self.width = words, int(height), int(width)

    @staticmethod
    def from_row_major_host(a, stream=None):
        """Upload a row-major numpy [height][width] array and transpose it on the GPU."""
        a = np.ascontiguousarray(a, dtype=np.uint32)
        h, w = a.shape
        src = to_device(a)
        dst = device_words(h * w)
        check(_L().sp1hip_transpose_to_col_major(_dptr(dst), _dptr(src), h, w, _stream_ptr(stream)))
        return ColMajor(dst, h, w)

    def to_row_major_host(self, stream=None):
        dst = device_words(self.height * self.width)
        check(_L().sp1hip_transpose_to_row_major(_dptr(dst), _dptr(self.words), self.height, self.width,
                                                 _stream_ptr(stream)))
        return to_host(dst, (self.height, self.width))

    def as_tensor_struct(self):
        return Tensor(C.c_void_p(self.words.data_ptr()), self.width)


def stage_tables(host_tables, stream=None):
    """Row-major host traces -> column-major device tables, one call for the whole shard (sp1hip_stage_tables:
    chunked PCIe copies on a side stream overlapped with the on-GPU transposes; mirrors `device_main_tracegen`'s
    "copy host trace to device", /root/reference/sp1-gpu/crates/jagged_tracegen/src/lib.rs:L819-L835).
    `host_tables`: 2-D numpy uint32 arrays or CPU torch int32 tensors (pinned for the asynchronous path), Montgomery
    words. Returns a list of ColMajor; each keeps its host source alive (the copies are asynchronous)."""
    from ._lib import HostTable
    n = len(host_tables)
    arr = (HostTable * max(n, 1))()
    outs = (C.c_void_p * max(n, 1))()
    result = []
    for i, a in enumerate(host_tables):
        if isinstance(a, np.ndarray):
            assert a.dtype == np.uint32 and a.ndim == 2 and a.flags["C_CONTIGUOUS"], "row-major uint32 [rows][cols]"
            ptr, (h, w) = a.ctypes.data, a.shape
        else:
            assert a.dtype == torch.int32 and a.dim() == 2 and a.is_contiguous() and not a.is_cuda
            ptr, (h, w) = a.data_ptr(), a.shape
        dst = device_words(h * w)
        cm = ColMajor(dst, h, w)
        cm._host_source = a
        arr[i] = HostTable(C.c_void_p(ptr), h, w)
        outs[i] = C.c_void_p(dst.data_ptr())
        result.append(cm)
    check(_L().sp1hip_stage_tables(arr, n, outs, _stream_ptr(stream)))
    return result


def _tensor_array(tensors):
    arr = (Tensor * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.as_tensor_struct()
    return arr


class DuplexChallenger:
    def __init__(self, handle=None):
        if handle is None:
            handle = C.c_void_p()
            check(_L().sp1hip_challenger_new(C.byref(handle)))
        self.h = handle

    def clone(self):
        out = C.c_void_p()
        check(_L().sp1hip_challenger_clone(self.h, C.byref(out)))
        return DuplexChallenger(out)

    def observe(self, felts):
        a = np.ascontiguousarray(np.asarray(felts, dtype=np.uint32).reshape(-1))
        check(_L().sp1hip_challenger_observe(self.h, a.ctypes.data_as(_lib.u32p), a.size))

    def sample(self):
        out = C.c_uint32()
        check(_L().sp1hip_challenger_sample(self.h, C.byref(out)))
        return out.value

    def sample_ext_element(self):
        e = Ext()
        check(_L().sp1hip_challenger_sample_ext(self.h, C.byref(e)))
        return np.array(list(e.c), dtype=np.uint32)

    def sample_point(self, n):
        return np.stack([self.sample_ext_element() for _ in range(n)]) if n else np.zeros((0, 4), np.uint32)

    def sample_bits(self, bits):
        out = C.c_uint32()
        check(_L().sp1hip_challenger_sample_bits(self.h, bits, C.byref(out)))
        return out.value

    def check_witness(self, bits, witness):
        ok = C.c_int()
        check(_L().sp1hip_challenger_check_witness(self.h, bits, C.c_uint32(int(witness)), C.byref(ok)))
        return bool(ok.value)

    def inject_pow_witnesses(self, witnesses):
        """Canonical proof-of-work witnesses for the next grinds, in order (sp1hip_challenger_inject_pow_witnesses)."""
        w = (C.c_uint32 * max(len(witnesses), 1))(*[int(x) for x in witnesses])
        check(_L().sp1hip_challenger_inject_pow_witnesses(self.h, w, len(witnesses)))

    def grind(self, bits, stream=None):
        out = C.c_uint32()
        check(_L().sp1hip_challenger_grind(self.h, bits, C.byref(out), _stream_ptr(stream)))
        return out.value

    def state(self):
        out = np.zeros(34, np.uint32)
        check(_L().sp1hip_challenger_state(self.h, out.ctypes.data_as(_lib.u32p)))
        return out

    def __del__(self):
        if getattr(self, "h", None) and _L is not None:      # _L is None during interpreter shutdown
            try:
                _L().sp1hip_challenger_free(self.h)
            except TypeError:                    # interpreter shutdown: the module globals are already gone
                pass
            self.h = None


class DftEncoder:
    """Reed–Solomon encoder: zero-pad by 2^log_blowup, forward DFT, bit-reversed rows."""

    def __init__(self, log_blowup):
        self.log_blowup = int(log_blowup)

    def encode_batch(self, mles, stream=None):
        out = []
        for m in mles:
            lg_n = m.height.bit_length() - 1
            assert 1 << lg_n == m.height
            cw = device_words((m.height << self.log_blowup) * m.width)
            check(_L().sp1hip_rs_encode_batch(_dptr(cw), _dptr(m.words), lg_n, self.log_blowup, m.width,
                                              _stream_ptr(stream)))
            out.append(ColMajor(cw, m.height << self.log_blowup, m.width))
        return out


class TcsProverData:
    def __init__(self, tree, root, commit, log_height, total_width):
        self.tree, self.root, self.commit = tree, root, commit
        self.log_height, self.total_width = log_height, total_width


class MerkleTcsProver:
    """Poseidon2KoalaBear16Prover: Merkle tensor commitment over rows of column-major tensors."""

    def commit_tensors(self, tensors, stream=None):
        h = tensors[0].height
        lg = h.bit_length() - 1
        assert all(t.height == h for t in tensors) and 1 << lg == h
        tree = device_words((2 * h - 1) * 8)
        rc = device_words(16)
        check(_L().sp1hip_merkle_commit(_tensor_array(tensors), len(tensors), lg, _dptr(tree), _dptr(rc),
                                        _stream_ptr(stream)))
        rc_h = to_host(rc)
        data = TcsProverData(tree, rc_h[:8].copy(), rc_h[8:].copy(), lg, sum(t.width for t in tensors))
        return data.commit, data

    def _indices(self, indices):
        return to_device(np.asarray(indices, dtype=np.uint32))

    def prove_openings_at_indices(self, data, indices, stream=None):
        idx = self._indices(indices)
        paths = device_words(len(indices) * data.log_height * 8)
        none = (Tensor * 1)(Tensor(None, 0))
        check(_L().sp1hip_merkle_open(none, 1, data.log_height, _dptr(data.tree), _dptr(idx), len(indices), None,
                                      _dptr(paths), _stream_ptr(stream)))
        return dict(merkle_root=data.root, log_tensor_height=data.log_height, width=data.total_width,
                    paths=to_host(paths, (len(indices), data.log_height, 8)))

    def compute_openings_at_indices(self, tensors, indices, stream=None):
        idx = self._indices(indices)
        tw = sum(t.width for t in tensors)
        lg = tensors[0].height.bit_length() - 1
        vals = device_words(len(indices) * tw)
        check(_L().sp1hip_merkle_open(_tensor_array(tensors), len(tensors), lg, None, _dptr(idx), len(indices),
                                      _dptr(vals), None, _stream_ptr(stream)))
        return to_host(vals, (len(indices), tw))


class BasefoldProverData:
    def __init__(self, handle, mles, commit):
        self.h, self.mles, self.commit = handle, mles, commit

    def codeword(self, k):
        ptr, width, lg = C.c_void_p(), C.c_uint32(), C.c_int()
        check(_L().sp1hip_basefold_data_codeword(self.h, k, C.byref(ptr), C.byref(width), C.byref(lg)))
        n = (1 << lg.value) * width.value
        out = device_words(n)
        check(_L().sp1hip_memcpy_d2d_async(_dptr(out), ptr, n * 4, _stream_ptr()))
        return ColMajor(out, 1 << lg.value, width.value)

    def tree(self):
        ptr, lg = C.c_void_p(), C.c_int()
        check(_L().sp1hip_basefold_data_tree(self.h, C.byref(ptr), C.byref(lg)))
        n = ((2 << lg.value) - 1) * 8
        out = device_words(n)
        check(_L().sp1hip_memcpy_d2d_async(_dptr(out), ptr, n * 4, _stream_ptr()))
        return to_host(out, (n // 8, 8))

    def __del__(self):
        if getattr(self, "h", None):
            try:
                _L().sp1hip_basefold_data_free(self.h)
            except TypeError:                    # interpreter shutdown: the module globals are already gone
                pass
            self.h = None


class BasefoldProver:
    def __init__(self, log_blowup=2, num_queries=124, proof_of_work_bits=16):
        self.config = FriConfig(log_blowup, num_queries, proof_of_work_bits)

    def commit_mles(self, mles, stream=None):
        lg_n = mles[0].height.bit_length() - 1
        assert all(m.height == 1 << lg_n for m in mles)
        commit = np.zeros(8, np.uint32)
        handle = C.c_void_p()
        check(_L().sp1hip_commit_mles(_tensor_array(mles), len(mles), lg_n, self.config.log_blowup,
                                      commit.ctypes.data_as(_lib.u32p), C.byref(handle), _stream_ptr(stream)))
        return commit, BasefoldProverData(handle, list(mles), commit)

    def evaluate_mles(self, mles, point, stream=None):
        """Claims for `prove_trusted_mle_evaluations`: every column of every mle at `point` ([dim][4])."""
        point = np.asarray(point, dtype=np.uint32).reshape(-1, 4)
        dim = point.shape[0]
        eq = device_words(4 << dim)
        check(_L().sp1hip_partial_lagrange(_ext_array(point), dim, _dptr(eq), _stream_ptr(stream)))
        tw = sum(m.width for m in mles)
        out = device_words(tw * 4)
        check(_L().sp1hip_mle_eval_columns(_tensor_array(mles), len(mles), dim, _dptr(eq), _dptr(out),
                                           _stream_ptr(stream)))
        return to_host(out, (tw, 4))

    def prove_trusted_mle_evaluations(self, eval_point, prover_data, evaluation_claims, challenger, stream=None):
        """prover_data: list of BasefoldProverData (one per commitment round); evaluation_claims: [total][4]."""
        point = np.asarray(eval_point, dtype=np.uint32).reshape(-1, 4)
        claims = np.asarray(evaluation_claims, dtype=np.uint32).reshape(-1, 4)
        widths = (C.c_uint32 * len(prover_data))(*[sum(m.width for m in pd.mles) for pd in prover_data])
        size = _L().sp1hip_basefold_proof_size(point.shape[0], widths, len(prover_data), self.config)
        buf = (C.c_uint8 * size)()
        n = C.c_size_t(size)
        handles = (C.c_void_p * len(prover_data))(*[pd.h for pd in prover_data])
        check(_L().sp1hip_basefold_prove(_ext_array(point), point.shape[0], handles, len(prover_data),
                                         _ext_array(claims), claims.shape[0], self.config, challenger.h, buf,
                                         C.byref(n), _stream_ptr(stream)))
        return C.string_at(buf, n.value)        # (slicing a c_uint8 array would build a list of ints first: ~20 ms per MB)


class _BorrowedBasefoldData:
    """Non-owning view of the BaseFold data inside a stacked commitment (freed with its owner)."""

    def __init__(self, handle, commit, widths):
        self.h, self.commit = handle, commit
        self.mles = [_Width(w) for w in widths]


class _Width:
    def __init__(self, width):
        self.width = width


class _RawColMajor:
    """A ColMajor-like view over device memory owned elsewhere."""

    def __init__(self, ptr, height, width):
        self.ptr, self.height, self.width = ptr, height, width

    def as_tensor_struct(self):
        return Tensor(self.ptr, self.width)


class StackedData:
    """Owns the dense stacked buffer + BaseFold data of one commitment (StackedBasefoldProverData)."""

    def __init__(self, handle, commit, num_added_vals, log_stacking_height):
        self.h, self.commit, self.num_added_vals, self.lsh = handle, commit, num_added_vals, log_stacking_height
        bf, nb, dense, padded = C.c_void_p(), C.c_int(), C.c_void_p(), C.c_uint64()
        check(_L().sp1hip_stacked_data_info(handle, C.byref(bf), C.byref(nb), C.byref(dense), C.byref(padded)))
        self.padded_area = padded.value
        self.batches = []
        for k in range(nb.value):
            t = Tensor()
            check(_L().sp1hip_stacked_batch(handle, k, C.byref(t)))
            self.batches.append(_RawColMajor(t.d_data, 1 << log_stacking_height, t.width))
        self.basefold = _BorrowedBasefoldData(bf, commit, [t.width for t in self.batches])

    def __del__(self):
        if getattr(self, "h", None):
            try:
                _L().sp1hip_stacked_data_free(self.h)
            except TypeError:                    # interpreter shutdown: the module globals are already gone
                pass
            self.h = None


def _table_array(tables):
    arr = (Table * max(len(tables), 1))()
    for i, t in enumerate(tables):
        arr[i] = Table(C.c_void_p(t.words.data_ptr()) if t.height * t.width else None, t.height, t.width)
    return arr


class StackedPcsProver:
    def __init__(self, log_stacking_height, batch_size, log_blowup=2):
        self.lsh, self.batch_size, self.log_blowup = log_stacking_height, batch_size, log_blowup

    def commit_multilinears(self, tables, stream=None):
        commit = np.zeros(8, np.uint32)
        added, handle = C.c_uint64(), C.c_void_p()
        check(_L().sp1hip_stacked_commit(_table_array(tables), len(tables), self.lsh, self.batch_size, self.log_blowup,
                                         commit.ctypes.data_as(_lib.u32p), C.byref(added), C.byref(handle),
                                         _stream_ptr(stream)))
        return commit, StackedData(handle, commit, added.value, self.lsh), added.value


class JaggedProver:
    """JaggedProver::commit_multilinears over chip tables (zero-row tables are counted, not committed)."""

    def __init__(self, max_log_row_count, log_stacking_height, batch_size, log_blowup=2):
        self.max_log_row_count, self.lsh = max_log_row_count, log_stacking_height
        self.batch_size, self.log_blowup = batch_size, log_blowup

    def commit_multilinears(self, tables, stream=None):
        commit = np.zeros(8, np.uint32)
        handle = C.c_void_p()
        check(_L().sp1hip_jagged_commit(_table_array(tables), len(tables), self.max_log_row_count, self.lsh,
                                        self.batch_size, self.log_blowup, commit.ctypes.data_as(_lib.u32p),
                                        C.byref(handle), _stream_ptr(stream)))
        return commit, StackedData(handle, commit, None, self.lsh)

    def prove_trusted_evaluations(self, z_row, claims_per_round, rounds, challenger, num_queries=124, pow_bits=16,
                                  stream=None):
        """JaggedProver::prove_trusted_evaluations (/root/reference/slop/crates/jagged/src/prover.rs:L162-L328).
        rounds: the StackedData of every commitment round, in order; claims_per_round[r]: [n_cols_r][4] column
        evaluations at z_row of round r's tables. Returns bincode(JaggedPcsProof); advances the challenger."""
        z_row = np.ascontiguousarray(np.asarray(z_row, dtype=np.uint32).reshape(-1, 4))
        assert z_row.shape[0] == self.max_log_row_count
        cl = [np.asarray(c, dtype=np.uint32).reshape(-1, 4) for c in claims_per_round]
        flat = np.ascontiguousarray(np.concatenate(cl)) if cl else np.zeros((0, 4), np.uint32)
        counts = (C.c_size_t * len(rounds))(*[c.shape[0] for c in cl])
        hs = (C.c_void_p * len(rounds))(*[r.h for r in rounds])
        cfg = FriConfig(self.log_blowup, num_queries, pow_bits)
        n = C.c_size_t(0)
        args = [_ext_array(z_row), self.max_log_row_count, hs, len(rounds), _ext_array(flat) if flat.size else None, counts,
                cfg, challenger.h]
        st = _L().sp1hip_jagged_prove(*args, None, C.byref(n), _stream_ptr(stream))
        if st != _lib.ERROR_BUFFER_TOO_SMALL:                                   # anything but BUFFER_TOO_SMALL is a real error
            check(st)
            raise RuntimeError("size query unexpectedly succeeded")
        buf = (C.c_uint8 * n.value)()
        check(_L().sp1hip_jagged_prove(*args, buf, C.byref(n), _stream_ptr(stream)))
        return C.string_at(buf, n.value)        # (slicing a c_uint8 array would build a list of ints first: ~20 ms per MB)


class ZerocheckChip:
    """One chip for `zerocheck`: an `sp1_amd.air.AirProgram`, column-major device traces (real rows only)."""

    def __init__(self, air, main, prep=None):
        self.air, self.main, self.prep = air, main, prep
        self.program = np.ascontiguousarray(air.to_array(), dtype=np.uint32)
        self.real_rows = main.height if main is not None else 0


def zerocheck(chips, max_log_row_count, zeta, openings, alpha, gkr_batch, public_values, challenger, stream=None):
    """ShardProver::zerocheck on the GPU. openings: per chip main then preprocessed column evaluations
    at zeta, concatenated [total][4]. Returns the proof bytes (layout in include/sp1hip.h)."""
    arr = (ZcChip * len(chips))()
    for i, c in enumerate(chips):
        arr[i] = ZcChip(c.program.ctypes.data_as(_lib.u32p), c.program.shape[0], c.air.main_width, c.air.prep_width,
                        c.air.num_constraints,
                        C.c_void_p(c.main.words.data_ptr()) if c.real_rows and c.air.main_width else None,
                        C.c_void_p(c.prep.words.data_ptr()) if c.real_rows and c.prep is not None else None, c.real_rows)
    pv = np.ascontiguousarray(np.asarray(public_values, dtype=np.uint32).reshape(-1))
    zeta = np.asarray(zeta, dtype=np.uint32).reshape(-1, 4)
    n = C.c_size_t(0)
    args = [arr, len(chips), max_log_row_count, _ext_array(zeta), _ext_array(openings), _ext(alpha), _ext(gkr_batch),
            pv.ctypes.data_as(_lib.u32p) if pv.size else None, int(pv.size), challenger.h]
    st = _L().sp1hip_zerocheck_prove(*args, None, C.byref(n), _stream_ptr(stream))
    if st != _lib.ERROR_BUFFER_TOO_SMALL:                                       # anything but BUFFER_TOO_SMALL is a real error
        check(st)
        raise RuntimeError("size query unexpectedly succeeded")
    buf = (C.c_uint8 * n.value)()
    check(_L().sp1hip_zerocheck_prove(*args, buf, C.byref(n), _stream_ptr(stream)))
    return C.string_at(buf, n.value)        # (slicing a c_uint8 array would build a list of ints first: ~20 ms per MB)


def logup_gkr(chips, max_log_row_count, challenger, stream=None):
    """GkrProverImpl::prove_logup_gkr on the GPU. chips: [(sp1_amd.air.InteractionProgram, main ColMajor or None,
    prep ColMajor or None)] in name order. Returns bincode(LogupGkrProof); advances the challenger."""
    arr = (GkrChip * len(chips))()
    keep = []
    for i, (prog, main, prep) in enumerate(chips):
        words = np.ascontiguousarray(prog.to_array(), dtype=np.uint32)
        keep.append(words)
        rows = main.height if main is not None else (prep.height if prep is not None else 0)
        arr[i] = GkrChip(prog.name.encode(), words.ctypes.data_as(_lib.u32p), words.size, prog.main_width, prog.prep_width,
                         C.c_void_p(main.words.data_ptr()) if main is not None and rows and prog.main_width else None,
                         C.c_void_p(prep.words.data_ptr()) if prep is not None and rows and prog.prep_width else None, rows)
    n = C.c_size_t(0)
    st = _L().sp1hip_logup_gkr_prove(arr, len(chips), max_log_row_count, challenger.h, None, C.byref(n), _stream_ptr(stream))
    if st != _lib.ERROR_BUFFER_TOO_SMALL:
        check(st)
        raise RuntimeError("size query unexpectedly succeeded")
    buf = (C.c_uint8 * n.value)()
    check(_L().sp1hip_logup_gkr_prove(arr, len(chips), max_log_row_count, challenger.h, buf, C.byref(n), _stream_ptr(stream)))
    return C.string_at(buf, n.value)        # (slicing a c_uint8 array would build a list of ints first: ~20 ms per MB)


def _shard_chip_array(chips):
    arr = (ShardChip * len(chips))()
    keep = []
    for i, (air, inter, main, prep) in enumerate(chips):
        prog = np.ascontiguousarray(air.to_array(), dtype=np.uint32)
        words = np.ascontiguousarray(inter.to_array(), dtype=np.uint32)
        keep += [prog, words]
        rows = main.height if main is not None else 0
        arr[i] = ShardChip(inter.name.encode(), prog.ctypes.data_as(_lib.u32p), prog.shape[0], air.num_constraints,
                           words.ctypes.data_as(_lib.u32p), words.size, air.main_width, air.prep_width,
                           C.c_void_p(main.words.data_ptr()) if rows else None,
                           C.c_void_p(prep.words.data_ptr()) if prep is not None and rows and air.prep_width else None, rows)
    return arr, keep


def prove_shard(chips, public_values, preprocessed, max_log_row_count, log_stacking_height, batch_size, challenger,
                log_blowup=2, num_queries=124, pow_bits=16, stream=None):
    """ShardProver::prove_shard_with_data on the GPU (sp1hip_prove_shard). chips: [(AirProgram, InteractionProgram, main
    ColMajor or None, prep ColMajor or None)] in name order; preprocessed: the StackedData of the preprocessed round
    (JaggedProver.commit_multilinears over the preprocessed traces). Returns bincode(ShardProof)."""
    arr, keep = _shard_chip_array(chips)
    pv = np.ascontiguousarray(np.asarray(public_values, dtype=np.uint32).reshape(-1))
    params = ShardParams(max_log_row_count, log_stacking_height, batch_size, FriConfig(log_blowup, num_queries, pow_bits))
    args = [arr, len(chips), pv.ctypes.data_as(_lib.u32p) if pv.size else None, int(pv.size), preprocessed.h, params, challenger.h]
    # One call in the steady state: the calling thread's buffer of its previous proof is offered first; a shard that needs
    # more is answered with ERROR_BUFFER_TOO_SMALL and the size before any work (and before the transcript is touched).
    import threading
    me = threading.get_ident()
    buf = _PROOF_BUFS.get(me)
    for _ in range(2):
        n = C.c_size_t(len(buf) if buf is not None else 0)
        st = _L().sp1hip_prove_shard(*args, buf, C.byref(n), _stream_ptr(stream))
        if st != _lib.ERROR_BUFFER_TOO_SMALL:
            check(st)
            return C.string_at(buf, n.value)        # (slicing a c_uint8 array would build a list of ints first: ~20 ms per MB)
        buf = _PROOF_BUFS[me] = (C.c_uint8 * n.value)()
    check(st)


_PROOF_BUFS = {}     # thread id -> proof buffer (prove_shard)


# main-trace widths and events per row of the recursion chips with device trace generation (sp1hip_tracegen_recursion_*)
RECURSION_TRACEGEN = {"BaseAlu": ("base_alu", 3, 3, 1), "ExtAlu": ("ext_alu", 12, 12, 1), "Select": ("select", 5, 5, 1),
                      "MemoryVar": ("memory_var", 8, 4, 2), "PrefixSumChecks": ("prefix_sum_checks", 15, 20, 1),
                      "Poseidon2WideDeg3": ("poseidon2_wide", 179, 32, 1)}


def tracegen_recursion(chip, events, height, stream=None):
    """Main trace of a recursion chip generated on the device from its event array (`generate_trace_device`).
    events: [n_events][event_words] uint32 Montgomery words (numpy, or an int32 device tensor). Returns a ColMajor."""
    fn, width, event_words, _ = RECURSION_TRACEGEN[chip]
    if isinstance(events, np.ndarray):
        events = to_device(np.ascontiguousarray(events, dtype=np.uint32).reshape(-1, event_words))
    n = events.numel() // event_words
    out = device_words(int(height) * width)
    check(getattr(_L(), "sp1hip_tracegen_recursion_" + fn)(_dptr(out), int(height), _dptr(events) if n else None, n, _stream_ptr(stream)))
    return ColMajor(out, int(height), width)


def tracegen_riscv_global(events, height, stream=None):
    """`generate_trace_device` for the RISC-V Global chip (sp1hip_tracegen_riscv_global): events = device int32 tensor [n, 9]
    (message[8], is_receive | kind << 8); returns the column-major [241][height] table as a ColMajor."""
    n = int(events.shape[0])
    out = device_words(241 * height)
    check(_L().sp1hip_tracegen_riscv_global(_dptr(out), int(height), _dptr(events) if n else None, n, _stream_ptr(stream)))
    return ColMajor(out, int(height), 241)


RISCV_ALU_CHIPS = {"Add": 0, "Addi": 1, "Sub": 2, "Addw": 3, "Subw": 4, "Mul": 5, "ShiftRight": 6, "Branch": 7}   # SP1HIP_RV64_CHIP_*
ALU_EVENT_WORDS = 11                                                                 # sp1hip_rv64_alu_event_t: 11 u64


def tracegen_riscv_alu(chip, events, height, stream=None):
    """`generate_trace_device` for one of the RISC-V instruction chips of RISCV_ALU_CHIPS (sp1hip_tracegen_riscv_alu): events = a
    device int64 tensor [n, 11] of sp1hip_rv64_alu_event_t records (riscv_exec.pack_alu_events); returns the column-major table
    [width][height] as a ColMajor."""
    kind = RISCV_ALU_CHIPS[chip]
    width = _L().sp1hip_tracegen_riscv_alu_width(kind)
    n = int(events.shape[0])
    assert events.dtype == torch.int64 and (n == 0 or (events.shape[1] == ALU_EVENT_WORDS and events.is_contiguous()))
    out = device_words(width * int(height))
    check(_L().sp1hip_tracegen_riscv_alu(kind, _dptr(out), int(height), _dptr(events) if n else None, n, _stream_ptr(stream)))
    return ColMajor(out, int(height), width)


class ProvingKey:
    """`ProvingKey` of the AirProver slot: the preprocessed commitment round + the verifying key (sp1hip_setup).
    Keeps the preprocessed device tables alive."""

    def __init__(self, prep_tables, max_log_row_count, log_stacking_height, batch_size, pc_start=(0, 0, 0),
                 initial_global_cumulative_sum=(0,) * 14, enable_untrusted_programs=0, log_blowup=2, num_queries=124, pow_bits=16,
                 stream=None):
        self.tables = list(prep_tables)
        self.params = ShardParams(max_log_row_count, log_stacking_height, batch_size, FriConfig(log_blowup, num_queries, pow_bits))
        arr = _table_array(self.tables)
        pc = (C.c_uint32 * 3)(*[int(x) for x in pc_start])
        cum = (C.c_uint32 * 14)(*[int(x) for x in initial_global_cumulative_sum])
        h = C.c_void_p()
        check(_L().sp1hip_setup(arr, len(self.tables), pc, cum, int(enable_untrusted_programs), self.params, C.byref(h),
                                _stream_ptr(stream)))
        self.h = h
        vk = _lib.Vk()
        check(_L().sp1hip_pk_vk(self.h, C.byref(vk)))
        self.vk = vk
        self.preprocessed_commit = np.array(list(vk.preprocessed_commit), dtype=np.uint32)

    def observe_into(self, challenger):
        """MachineVerifyingKey::observe_into."""
        check(_L().sp1hip_vk_observe_into(C.byref(self.vk), challenger.h))

    def prove_shard(self, chips, public_values, pow_witnesses=(), stream=None):
        """AirProver::prove_shard_with_pk from the generated traces on -> bincode(ShardProof)."""
        arr, keep = _shard_chip_array(chips)
        pv = np.ascontiguousarray(np.asarray(public_values, dtype=np.uint32).reshape(-1))
        w = (C.c_uint32 * max(len(pow_witnesses), 1))(*[int(x) for x in pow_witnesses])
        args = [self.h, arr, len(chips), pv.ctypes.data_as(_lib.u32p) if pv.size else None, int(pv.size), w, len(pow_witnesses)]
        # one call in the steady state: the buffer of the previous proof of this key is offered first (the size depends on
        # the shard's shape only); a shard that needs more answers ERROR_BUFFER_TOO_SMALL with the size, before any work
        import threading
        bufs = self.__dict__.setdefault("_proof_bufs", {})            # per calling thread: two threads may prove with one key
        me = threading.get_ident()
        buf = bufs.get(me)
        for _ in range(2):
            n = C.c_size_t(len(buf) if buf is not None else 0)
            st = _L().sp1hip_prove_shard_with_pk(*args, buf, C.byref(n), _stream_ptr(stream))
            if st != _lib.ERROR_BUFFER_TOO_SMALL:
                check(st)
                return C.string_at(buf, n.value)
            buf = bufs[me] = (C.c_uint8 * n.value)()
        check(st)

    def __del__(self):
        if getattr(self, "h", None):
            try:
                _L().sp1hip_pk_free(self.h)
            except TypeError:                    # interpreter shutdown: the module globals are already gone
                pass
            self.h = None


def bb_commit_mles(mles, log_blowup, stream=None):
    """BabyBear `commit_mles` (sp1hip_bb_commit_mles): mles = ColMajor tensors of equal height 2^lg_n holding BabyBear Montgomery
    words. Returns (commitment [8] numpy, codewords [ColMajor], tree words as a device tensor [(2 N - 1) * 8])."""
    lg_n = mles[0].height.bit_length() - 1
    N = mles[0].height << log_blowup
    arr = (Tensor * len(mles))(*[m.as_tensor_struct() for m in mles])
    cws = [device_words(N * m.width) for m in mles]
    ptrs = (C.c_void_p * len(mles))(*[c.data_ptr() for c in cws])
    tree = device_words((2 * N - 1) * 8)
    commit = np.zeros(8, np.uint32)
    check(_L().sp1hip_bb_commit_mles(arr, len(mles), lg_n, log_blowup, ptrs, _dptr(tree), commit.ctypes.data_as(_lib.u32p), _stream_ptr(stream)))
    return commit, [ColMajor(c, N, m.width) for c, m in zip(cws, mles)], tree


def bb_poseidon2_permute(states, stream=None):
    """[n, 16] numpy BabyBear Montgomery words -> permuted (sp1hip_bb_poseidon2_permute)."""
    d = to_device(np.ascontiguousarray(states, dtype=np.uint32).reshape(-1))
    check(_L().sp1hip_bb_poseidon2_permute(_dptr(d), d.numel() // 16, _stream_ptr(stream)))
    return to_host(d, (d.numel() // 16, 16))


class PinnedHost:
    """Pinned host words (sp1hip_malloc_host) holding one row-major table: what a host trace generator fills and
    `ProverPool.submit` uploads at full PCIe rate."""

    def __init__(self, table):
        table = np.ascontiguousarray(table, dtype=np.uint32)
        self.shape = table.shape
        p = C.c_void_p()
        check(_L().sp1hip_malloc_host(C.byref(p), max(table.nbytes, 4)))
        self.ptr = p
        if table.nbytes:
            C.memmove(p, table.ctypes.data, table.nbytes)

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                _L().sp1hip_free_host(self.ptr)
            except TypeError:                    # interpreter shutdown: the module globals are already gone
                pass
            self.ptr = None


class ProverPool:
    """N shard proofs in flight on one GPU (sp1hip_pool_*): `n_slots` prover slots (thread + stream each) plus a stager
    that uploads host traces of the next shards while the slots prove — the library-side `ProverSemaphore`."""

    def __init__(self, n_slots, device=None):
        import torch
        h = C.c_void_p()
        check(_L().sp1hip_pool_create(torch.cuda.current_device() if device is None else int(device), int(n_slots), C.byref(h)))
        self.h, self.n_slots, self._keep = h, int(n_slots), {}

    def submit(self, pk, chips, public_values=()):
        """chips: [(AirProgram, InteractionProgram, main, prep)] in name order; main is a ColMajor (resident in HBM), a
        PinnedHost (row-major host words the pool stages) or None; prep: the ColMajor the proving key was set up from."""
        arr = (_lib.PoolChip * len(chips))()
        keep = [chips, pk]
        for i, (air, inter, main, prep) in enumerate(chips):
            prog = np.ascontiguousarray(air.to_array(), dtype=np.uint32)
            words = np.ascontiguousarray(inter.to_array(), dtype=np.uint32)
            keep += [prog, words]
            if isinstance(main, PinnedHost):
                rows, h_main, d_main = main.shape[0], main.ptr, None
            else:
                rows = main.height if main is not None else 0
                h_main, d_main = None, (C.c_void_p(main.words.data_ptr()) if rows else None)
            arr[i] = _lib.PoolChip(inter.name.encode(), prog.ctypes.data_as(_lib.u32p), prog.shape[0], air.num_constraints,
                                   words.ctypes.data_as(_lib.u32p), words.size, air.main_width, air.prep_width, h_main, d_main,
                                   C.c_void_p(prep.words.data_ptr()) if prep is not None and rows and air.prep_width else None, rows)
        pv = np.ascontiguousarray(np.asarray(public_values, dtype=np.uint32).reshape(-1))
        keep += [arr, pv]
        t = C.c_uint64()
        check(_L().sp1hip_pool_submit(self.h, pk.h, arr, len(chips), pv.ctypes.data_as(_lib.u32p) if pv.size else None, int(pv.size),
                                      C.byref(t)))
        self._keep[t.value] = keep
        return t.value

    def wait(self, ticket, block=True):
        """-> (bincode(ShardProof), {"staging_ms", "queued_ms", "proving_ms", "slot"}); block=False returns None while the
        shard is in flight."""
        fn = _L().sp1hip_pool_wait if block else _L().sp1hip_pool_try_wait
        n, times = C.c_size_t(0), _lib.PoolTimes()
        st = fn(self.h, ticket, None, C.byref(n), C.byref(times))
        if st == _lib.ERROR_NOT_READY:
            return None
        if st != _lib.ERROR_BUFFER_TOO_SMALL:
            self._keep.pop(ticket, None)
            check(st)
            raise RuntimeError("size query unexpectedly succeeded")
        buf = (C.c_uint8 * n.value)()
        st = fn(self.h, ticket, buf, C.byref(n), C.byref(times))
        self._keep.pop(ticket, None)
        check(st)
        return C.string_at(buf, n.value), {"staging_ms": times.staging_ms, "queued_ms": times.queued_ms,
                                           "proving_ms": times.proving_ms, "slot": times.slot}

    def close(self):
        if getattr(self, "h", None):
            _L().sp1hip_pool_destroy(self.h)
            self.h = None
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except TypeError:                        # interpreter shutdown: the module globals are already gone
            pass


def parse_logup_gkr_proof(blob):
    """The `logup_evaluations` of a bincode(LogupGkrProof): (point [L][4], [(name, main [w][4], prep [w][4] or None)])
    in Montgomery words — the zeta / openings that zerocheck consumes."""
    import struct
    o = 0

    def u64():
        nonlocal o
        v = struct.unpack_from("<Q", blob, o)[0]
        o += 8
        return v

    def exts(k):
        nonlocal o
        a = np.frombuffer(blob, dtype="<u4", count=4 * k, offset=o).astype(np.uint64)
        o += 16 * k
        return ((a << np.uint64(32)) % np.uint64(P)).astype(np.uint32).reshape(k, 4)

    for _ in range(2):                                     # circuit output
        exts(u64())
        o += 24
    for _ in range(u64()):                                 # round proofs
        exts(4)
        for _ in range(u64()):
            exts(u64())
        exts(1)
        exts(u64())
        exts(1)
    point = exts(u64())
    chips = []
    for _ in range(u64()):
        n = u64()
        name = blob[o:o + n].decode()
        o += n
        main = exts(u64())
        o += 16
        prep = None
        o += 1
        if blob[o - 1]:
            prep = exts(u64())
            o += 16
        chips.append((name, main, prep))
    assert o + 4 == len(blob)
    return point, chips


def parse_zerocheck_proof(blob):
    """Split the zerocheck output (layout in include/sp1hip.h) into the sumcheck point (Montgomery words,
    [dim][4]) and the per-chip opened values (one [w_prep + w_main][4] array per chip: preprocessed columns
    first, then main) — what the jagged evaluation proof takes as z_row and as its column claims."""
    import struct
    P_ = P
    o = 0

    def u64():
        nonlocal o
        v = struct.unpack_from("<Q", blob, o)[0]
        o += 8
        return v

    def exts(k):
        nonlocal o
        a = np.frombuffer(blob, dtype="<u4", count=4 * k, offset=o).astype(np.uint64)
        o += 16 * k
        return ((a << np.uint64(32)) % np.uint64(P_)).astype(np.uint32).reshape(k, 4)    # canonical -> Montgomery

    for _ in range(u64()):
        exts(u64())
    exts(1)
    point = exts(u64())
    exts(1)
    chips = []
    for _ in range(u64()):
        chips.append(exts(u64()))
    assert o == len(blob)
    return point, chips
