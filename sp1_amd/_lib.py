"""ctypes binding of libsp1hip.so (the C-ABI drop-in boundary declared in include/sp1hip.h).

This module only loads the in-tree shared library and declares prototypes. There is NO fallback:
if the HIP extension is missing or a call fails, an exception is raised (the product path must fail
loudly, never route through a CPU implementation).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsp1hip.so")

u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class Ext(C.Structure):
    _fields_ = [("c", C.c_uint32 * 4)]


class Tensor(C.Structure):
    _fields_ = [("d_data", C.c_void_p), ("width", C.c_uint32)]


class Table(C.Structure):
    _fields_ = [("d_data", C.c_void_p), ("rows", C.c_uint64), ("cols", C.c_uint32)]


class HostTable(C.Structure):
    _fields_ = [("h_data", C.c_void_p), ("rows", C.c_uint64), ("cols", C.c_uint32)]


class ZcChip(C.Structure):
    _fields_ = [("program", C.POINTER(C.c_uint32)), ("n_instr", C.c_uint32), ("main_width", C.c_uint32),
                ("prep_width", C.c_uint32), ("num_constraints", C.c_uint32), ("d_main", C.c_void_p),
                ("d_prep", C.c_void_p), ("real_rows", C.c_uint64)]


class GkrChip(C.Structure):
    _fields_ = [("name", C.c_char_p), ("interactions", C.POINTER(C.c_uint32)), ("n_words", C.c_uint32),
                ("main_width", C.c_uint32), ("prep_width", C.c_uint32), ("d_main", C.c_void_p), ("d_prep", C.c_void_p),
                ("real_rows", C.c_uint64)]


class FriConfig(C.Structure):
    _fields_ = [("log_blowup", C.c_int), ("num_queries", C.c_int), ("proof_of_work_bits", C.c_int)]


class ShardChip(C.Structure):
    _fields_ = [("name", C.c_char_p), ("program", C.POINTER(C.c_uint32)), ("n_instr", C.c_uint32),
                ("num_constraints", C.c_uint32), ("interactions", C.POINTER(C.c_uint32)), ("n_words", C.c_uint32),
                ("main_width", C.c_uint32), ("prep_width", C.c_uint32), ("d_main", C.c_void_p), ("d_prep", C.c_void_p),
                ("real_rows", C.c_uint64)]


class PoolChip(C.Structure):
    _fields_ = [("name", C.c_char_p), ("program", C.POINTER(C.c_uint32)), ("n_instr", C.c_uint32),
                ("num_constraints", C.c_uint32), ("interactions", C.POINTER(C.c_uint32)), ("n_words", C.c_uint32),
                ("main_width", C.c_uint32), ("prep_width", C.c_uint32), ("h_main", C.c_void_p), ("d_main", C.c_void_p),
                ("d_prep", C.c_void_p), ("real_rows", C.c_uint64)]


class PoolTimes(C.Structure):
    _fields_ = [("staging_ms", C.c_double), ("queued_ms", C.c_double), ("proving_ms", C.c_double), ("slot", C.c_int)]


class ShardParams(C.Structure):
    _fields_ = [("max_log_row_count", C.c_int), ("log_stacking_height", C.c_int), ("batch_size", C.c_int),
                ("fri", FriConfig)]


class Rv64ShardInfo(C.Structure):
    _fields_ = [("shard", C.c_uint64), ("n_cycles", C.c_uint64), ("n_events", C.c_uint64), ("n_local", C.c_uint64), ("n_keccak", C.c_uint64), ("n_poseidon2", C.c_uint64), ("n_sha_extend", C.c_uint64), ("n_sha_compress", C.c_uint64), ("n_uint256", C.c_uint64), ("n_secp256k1_add", C.c_uint64), ("n_secp256k1_double", C.c_uint64),
                ("pc_start", C.c_uint64), ("next_pc", C.c_uint64), ("clk_start", C.c_uint64), ("clk_end", C.c_uint64),
                ("halted", C.c_uint32), ("exit_code", C.c_uint32), ("commit_syscall", C.c_uint32),
                ("commit_deferred_syscall", C.c_uint32), ("committed_value_digest", C.c_uint32 * 8),
                ("deferred_proofs_digest", C.c_uint32 * 8), ("estimated_area", C.c_uint64), ("estimated_max_height", C.c_uint64)]


class Rv64ShardLimits(C.Structure):
    _fields_ = [("element_threshold", C.c_uint64), ("height_threshold", C.c_uint64), ("fixed_area", C.c_uint64),
                ("opcode_cost", C.c_uint64 * 64), ("opcode_chip", C.c_uint32 * 64), ("alu_x0_cost", C.c_uint64), ("load_x0_cost", C.c_uint64),
                ("memory_local_cost", C.c_uint64), ("global_cost", C.c_uint64), ("syscall_core_cost", C.c_uint64),
                ("memory_bump_cost", C.c_uint64), ("state_bump_cost", C.c_uint64)]


class Rv64AluEvent(C.Structure):
    """sp1hip_rv64_alu_event_t: what sp1hip_tracegen_riscv_alu makes a row of (riscv_exec.pack_alu_events builds arrays of them)."""
    _fields_ = [(n, C.c_uint64) for n in ("pc", "clk", "ops", "a", "b", "c", "a_prev", "a_pts", "b_pts", "c_pts", "aux")]


class Vk(C.Structure):
    _fields_ = [("pc_start", C.c_uint32 * 3), ("initial_global_cumulative_sum", C.c_uint32 * 14),
                ("preprocessed_commit", C.c_uint32 * 8), ("enable_untrusted_programs", C.c_uint32)]


# status codes of include/sp1hip.h
SUCCESS, ERROR_INVALID_ARGUMENT, ERROR_NOT_READY, ERROR_BUFFER_TOO_SMALL = 0, -1, -3, -6


class Sp1HipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("sp1hip status %d: %s" % (status, message))
        self.status = status


# every symbol include/sp1hip.h declares: (name, restype, argtypes); restype None => int status
_vp = C.c_void_p
_sz = C.c_size_t
_int = C.c_int
PROTOTYPES = [
    ("sp1hip_last_error", C.c_char_p, []),
    ("sp1hip_version", C.c_char_p, []),
    ("sp1hip_device_count", None, [C.POINTER(_int)]),
    ("sp1hip_set_device", None, [_int]),
    ("sp1hip_get_device", None, [C.POINTER(_int)]),
    ("sp1hip_mem_info", None, [C.POINTER(_sz), C.POINTER(_sz)]),
    ("sp1hip_mem_trim", None, [C.POINTER(_sz)]),
    ("sp1hip_stream_release", None, [C.c_void_p]),
    ("sp1hip_host_threads", _int, []),
    ("sp1hip_malloc", None, [C.POINTER(_vp), _sz]),
    ("sp1hip_free", None, [_vp]),
    ("sp1hip_malloc_async", None, [C.POINTER(_vp), _sz, _vp]),
    ("sp1hip_free_async", None, [_vp, _vp]),
    ("sp1hip_malloc_host", None, [C.POINTER(_vp), _sz]),
    ("sp1hip_free_host", None, [_vp]),
    ("sp1hip_memcpy_h2d_async", None, [_vp, _vp, _sz, _vp]),
    ("sp1hip_memcpy_d2h_async", None, [_vp, _vp, _sz, _vp]),
    ("sp1hip_memcpy_d2d_async", None, [_vp, _vp, _sz, _vp]),
    ("sp1hip_memset_async", None, [_vp, _int, _sz, _vp]),
    ("sp1hip_stream_create", None, [C.POINTER(_vp)]),
    ("sp1hip_stream_destroy", None, [_vp]),
    ("sp1hip_stream_synchronize", None, [_vp]),
    ("sp1hip_stream_query", None, [_vp]),
    ("sp1hip_event_create", None, [C.POINTER(_vp)]),
    ("sp1hip_event_destroy", None, [_vp]),
    ("sp1hip_event_record", None, [_vp, _vp]),
    ("sp1hip_event_synchronize", None, [_vp]),
    ("sp1hip_event_elapsed_ms", None, [C.POINTER(C.c_float), _vp, _vp]),
    ("sp1hip_stream_wait_event", None, [_vp, _vp]),
    ("sp1hip_timers_enable", None, [_int]),
    ("sp1hip_timers_reset", None, []),
    ("sp1hip_timers_read", None, [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    ("sp1hip_transpose_to_col_major", None, [_vp, _vp, _sz, _sz, _vp]),
    ("sp1hip_transpose_to_row_major", None, [_vp, _vp, _sz, _sz, _vp]),
    ("sp1hip_stage_tables", None, [C.POINTER(HostTable), _int, C.POINTER(_vp), _vp]),
    ("sp1hip_host_register", None, [_vp, _sz]),
    ("sp1hip_host_unregister", None, [_vp]),
    ("sp1hip_to_monty", None, [_vp, _sz, _vp]),
    ("sp1hip_from_monty", None, [_vp, _sz, _vp]),
    ("sp1hip_rs_encode_batch", None, [_vp, _vp, _int, _int, _sz, _vp]),
    ("sp1hip_merkle_commit", None, [C.POINTER(Tensor), _int, _int, _vp, _vp, _vp]),
    ("sp1hip_merkle_open", None, [C.POINTER(Tensor), _int, _int, _vp, _vp, _sz, _vp, _vp, _vp]),
    ("sp1hip_poseidon2_permute", None, [_vp, _sz, _vp]),
    ("sp1hip_bb_rs_encode_batch", None, [_vp, _vp, _int, _int, _sz, _vp]),
    ("sp1hip_bb_merkle_commit", None, [C.POINTER(Tensor), _int, _int, _vp, _vp, _vp]),
    ("sp1hip_bb_commit_mles", None, [C.POINTER(Tensor), _int, _int, _int, C.POINTER(_vp), _vp, u32p, _vp]),
    ("sp1hip_bb_poseidon2_permute", None, [_vp, _sz, _vp]),
    ("sp1hip_poseidon2_permute_integer_form", None, [_vp, _sz, _vp]),
    ("sp1hip_poseidon2_permute_host", None, [_vp, _sz, _int]),
    ("sp1hip_host_permutation_is_vectorised", None, []),
    ("sp1hip_gkr_host_simd_available", None, []),
    ("sp1hip_gkr_host_round_sums", None, [_vp, _sz, _vp, _sz, _sz, _vp]),
    ("sp1hip_gkr_host_round_fold", None, [_vp, _vp, _sz, _sz, _vp]),
    ("sp1hip_basefold_batch", None, [C.POINTER(Tensor), _int, _int, _vp, _vp, _vp]),
    ("sp1hip_fold_even_odd", None, [_vp, _int, Ext, _vp, _vp]),
    ("sp1hip_fold_mle", None, [_vp, _int, Ext, _vp, _vp]),
    ("sp1hip_partial_lagrange", None, [C.POINTER(Ext), _int, _vp, _vp]),
    ("sp1hip_mle_eval_columns", None, [C.POINTER(Tensor), _int, _int, _vp, _vp, _vp]),
    ("sp1hip_fix_last_variable", None, [_vp, C.c_uint64, C.c_uint32, _int, Ext, _vp, _vp, _vp]),
    ("sp1hip_ext_fixed_at_zero", None, [_vp, _int, _vp, _vp, _vp]),
    ("sp1hip_challenger_new", None, [C.POINTER(_vp)]),
    ("sp1hip_challenger_clone", None, [_vp, C.POINTER(_vp)]),
    ("sp1hip_challenger_free", "void", [_vp]),
    ("sp1hip_challenger_observe", None, [_vp, u32p, _sz]),
    ("sp1hip_challenger_sample", None, [_vp, u32p]),
    ("sp1hip_challenger_sample_ext", None, [_vp, C.POINTER(Ext)]),
    ("sp1hip_challenger_sample_bits", None, [_vp, _int, u32p]),
    ("sp1hip_challenger_check_witness", None, [_vp, _int, C.c_uint32, C.POINTER(_int)]),
    ("sp1hip_challenger_grind", None, [_vp, _int, u32p, _vp]),
    ("sp1hip_challenger_inject_pow_witnesses", None, [_vp, u32p, _int]),
    ("sp1hip_challenger_state", None, [_vp, u32p]),
    ("sp1hip_commit_mles", None, [C.POINTER(Tensor), _int, _int, _int, u32p, C.POINTER(_vp), _vp]),
    ("sp1hip_basefold_data_free", "void", [_vp]),
    ("sp1hip_basefold_data_codeword", None, [_vp, _int, C.POINTER(_vp), u32p, C.POINTER(_int)]),
    ("sp1hip_basefold_data_tree", None, [_vp, C.POINTER(_vp), C.POINTER(_int)]),
    ("sp1hip_basefold_prove", None, [C.POINTER(Ext), _int, C.POINTER(_vp), _int, C.POINTER(Ext), _sz, FriConfig, _vp,
                                     u8p, C.POINTER(_sz), _vp]),
    ("sp1hip_basefold_proof_size", _sz, [_int, u32p, _int, FriConfig]),
    ("sp1hip_stacked_commit", None, [C.POINTER(Table), _int, _int, _int, _int, u32p, C.POINTER(C.c_uint64),
                                     C.POINTER(_vp), _vp]),
    ("sp1hip_stacked_data_free", "void", [_vp]),
    ("sp1hip_stacked_data_info", None, [_vp, C.POINTER(_vp), C.POINTER(_int), C.POINTER(_vp), C.POINTER(C.c_uint64)]),
    ("sp1hip_stacked_batch", None, [_vp, _int, C.POINTER(Tensor)]),
    ("sp1hip_jagged_commit", None, [C.POINTER(Table), _int, _int, _int, _int, _int, u32p, C.POINTER(_vp), _vp]),
    ("sp1hip_jagged_prove", None, [C.POINTER(Ext), _int, C.POINTER(_vp), _int, C.POINTER(Ext), C.POINTER(_sz), FriConfig, _vp,
                                   u8p, C.POINTER(_sz), _vp]),
    ("sp1hip_logup_gkr_prove", None, [C.POINTER(GkrChip), _int, _int, _vp, u8p, C.POINTER(_sz), _vp]),
    ("sp1hip_prove_shard", None, [C.POINTER(ShardChip), _int, u32p, _int, _vp, ShardParams, _vp, u8p, C.POINTER(_sz), _vp]),
    ("sp1hip_tracegen_recursion_base_alu", None, [_vp, C.c_uint64, _vp, C.c_uint64, _vp]),
    ("sp1hip_tracegen_recursion_ext_alu", None, [_vp, C.c_uint64, _vp, C.c_uint64, _vp]),
    ("sp1hip_tracegen_recursion_select", None, [_vp, C.c_uint64, _vp, C.c_uint64, _vp]),
    ("sp1hip_tracegen_recursion_memory_var", None, [_vp, C.c_uint64, _vp, C.c_uint64, _vp]),
    ("sp1hip_tracegen_recursion_prefix_sum_checks", None, [_vp, C.c_uint64, _vp, C.c_uint64, _vp]),
    ("sp1hip_tracegen_recursion_poseidon2_wide", None, [_vp, C.c_uint64, _vp, C.c_uint64, _vp]),
    ("sp1hip_tracegen_riscv_global", None, [_vp, C.c_uint64, _vp, C.c_uint64, _vp]),
    ("sp1hip_tracegen_riscv_alu_width", None, [C.c_int]),
    ("sp1hip_tracegen_riscv_alu", None, [C.c_int, _vp, C.c_uint32, _vp, C.c_uint32, _vp]),
    ("sp1hip_rv64_create", None, [u8p, C.c_uint64, C.POINTER(_vp)]),
    ("sp1hip_rv64_destroy", "void", [_vp]),
    ("sp1hip_rv64_write_stdin", None, [_vp, u8p, C.c_uint64]),
    ("sp1hip_rv64_run_shard", None, [_vp, C.c_uint64, C.POINTER(Rv64ShardInfo)]),
    ("sp1hip_rv64_set_recording", None, [_vp, _int]),
    ("sp1hip_rv64_set_shard_limits", None, [_vp, C.POINTER(Rv64ShardLimits)]),
    ("sp1hip_rv64_events", C.POINTER(C.c_uint64), [_vp]),
    ("sp1hip_rv64_local_memory", C.POINTER(C.c_uint64), [_vp]),
    ("sp1hip_rv64_keccak_events", C.POINTER(C.c_uint64), [_vp]),
    ("sp1hip_rv64_poseidon2_events", C.POINTER(C.c_uint64), [_vp]),
    ("sp1hip_rv64_sha_extend_events", C.POINTER(C.c_uint64), [_vp]),
    ("sp1hip_rv64_sha_compress_events", C.POINTER(C.c_uint64), [_vp]),
    ("sp1hip_rv64_uint256_events", C.POINTER(C.c_uint64), [_vp]),
    ("sp1hip_rv64_secp256k1_add_events", C.POINTER(C.c_uint64), [_vp]),
    ("sp1hip_rv64_secp256k1_double_events", C.POINTER(C.c_uint64), [_vp]),
    ("sp1hip_rv64_precompile_events", None, [_vp, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint64))]),
    ("sp1hip_rv64_program", None, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint64))]),
    ("sp1hip_rv64_global_memory", None, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint64))]),
    ("sp1hip_rv64_memory_image", None, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint64))]),
    ("sp1hip_rv64_output", None, [_vp, _int, C.POINTER(u8p), C.POINTER(C.c_uint64)]),
    ("sp1hip_setup", None, [C.POINTER(Table), _int, u32p, u32p, C.c_uint32, ShardParams, C.POINTER(_vp), _vp]),
    ("sp1hip_pk_free", "void", [_vp]),
    ("sp1hip_pk_vk", None, [_vp, C.POINTER(Vk)]),
    ("sp1hip_vk_observe_into", None, [C.POINTER(Vk), _vp]),
    ("sp1hip_prove_shard_with_pk", None, [_vp, C.POINTER(ShardChip), _int, u32p, _int, u32p, _int, u8p, C.POINTER(_sz), _vp]),
    ("sp1hip_zerocheck_prove", None, [C.POINTER(ZcChip), _int, _int, C.POINTER(Ext), C.POINTER(Ext), Ext, Ext, u32p, _int,
                                      _vp, u8p, C.POINTER(_sz), _vp]),
    ("sp1hip_pool_create", None, [_int, _int, C.POINTER(_vp)]),
    ("sp1hip_pool_destroy", "void", [_vp]),
    ("sp1hip_pool_submit", None, [_vp, _vp, C.POINTER(PoolChip), _int, u32p, _int, C.POINTER(C.c_uint64)]),
    ("sp1hip_pool_wait", None, [_vp, C.c_uint64, u8p, C.POINTER(_sz), C.POINTER(PoolTimes)]),
    ("sp1hip_pool_try_wait", None, [_vp, C.c_uint64, u8p, C.POINTER(_sz), C.POINTER(PoolTimes)]),
    ("sp1hip_zerocheck_biv_interp_host", None, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p]),
    ("sp1hip_zerocheck_plan_eval", None, [u32p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p, u32p, C.c_uint32, _int, u32p,
                                          C.c_uint32, u32p]),
    ("sp1hip_zerocheck_poly_check", None, [u32p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, Ext, C.POINTER(Ext), C.POINTER(Ext), u32p]),
]

_lib = None


def load():
    """Load libsp1hip.so and declare every prototype. Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # If torch is going to be used in this process it must be imported BEFORE libsp1hip.so: torch ships
    # its own libamdhip64.so and two HIP runtimes in one process do not share devices or streams.
    # Importing torch first makes the dynamic linker bind this library to the runtime torch loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, res, args in PROTOTYPES:
        fn = getattr(lib, name)  # AttributeError if the ABI symbol is missing
        fn.argtypes = args
        if res is None:
            fn.restype = C.c_int
        elif res == "void":
            fn.restype = None
        else:
            fn.restype = res
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise Sp1HipError(status, load().sp1hip_last_error().decode(errors="replace"))
