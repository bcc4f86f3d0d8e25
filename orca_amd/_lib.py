"""ctypes binding of liborca_hip.so (include/orca_hip.h).  There is no CPU fallback: if the
library is missing or a call fails, this raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ORCA_HIP_LIB: another build of the same library, for A/B measurements of two kernels in one process environment)
LIB_PATH = os.environ.get("ORCA_HIP_LIB") or os.path.join(_HERE, "liborca_hip.so")

ORCA_D = 100
ORCA_MAX_DE = 192
ORCA_MAX_K = 20
MAX_WORLD = 64      # ORCA_MAX_WORLD

vp = ctypes.c_void_p
i64 = ctypes.c_int64
i32 = ctypes.c_int
u64 = ctypes.c_uint64
u8 = ctypes.c_uint8


class TgatInputs(ctypes.Structure):
  """orca_tgat_inputs (include/orca_hip.h)."""
  _fields_ = [(n, vp) for n in (
      "src_table", "src_idx", "src_overlay", "src_flags",
      "nbr_table", "nbr_idx", "nbr_overlay", "nbr_flags",
      "nbr_ids", "mask", "edge_table", "edge_idx",
      "dt", "time_w", "time_b", "src_time", "nbr_time")] + [("q_folded", ctypes.c_int32), ("dropout_p", ctypes.c_float),
                                                                     ("dropout_seed", ctypes.c_uint64), ("dropout_q0", ctypes.c_int64)]


class TgatParams(ctypes.Structure):
  """orca_tgat_params (include/orca_hip.h): raw state_dict tensors, in ops.TGAT_KEYS order."""
  _fields_ = [(n, vp) for n in ("q_w", "k_w", "v_w", "in_b", "o_w", "o_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                "ln_w", "ln_b")]


ORCA_ENGINE_MAX_LAYERS = 4


class SlotCache(ctypes.Structure):
  """orca_slot_cache (include/orca_hip.h): the budget-bounded history of the native Path C step."""
  _fields_ = [(n, vp) for n in ("rows", "stamp", "prev_plan", "pull_idx")] + \
             [("plan_cap", i64), ("q_bound", i64), ("gen", i64), ("prev_n", i64), ("cur", ctypes.c_int32)]


class EngineDesc(ctypes.Structure):
  """orca_engine_desc (include/orca_hip.h)."""
  _fields_ = ([("n_nodes", i64), ("n_layers", ctypes.c_int32), ("K", ctypes.c_int32), ("d_e", ctypes.c_int32),
               ("small_batch", ctypes.c_int32), ("csr", vp), ("edge_feats", vp), ("time_w", vp), ("time_b", vp),
               ("att_blob", vp * ORCA_ENGINE_MAX_LAYERS), ("gru_blob", vp), ("memory", vp), ("last_update", vp),
               ("messages", vp), ("msg_ts", vp), ("pending", vp), ("overlay", vp),
               ("hist", vp * (ORCA_ENGINE_MAX_LAYERS - 1)), ("slot", vp), ("worklist", vp), ("count", vp),
               ("q_max", i64), ("nodes", vp), ("qts", vp), ("nodes32", vp), ("nbr", vp), ("eid", vp), ("et", vp),
               ("dt", vp), ("emb", vp * 2), ("winners", vp), ("eager_ids", vp), ("eager_count", vp),
               ("att_q_folded", ctypes.c_int32), ("world", ctypes.c_int32), ("rank", ctypes.c_int32),
               ("xm", vp), ("xm_own", vp), ("xl", vp), ("xl_own", vp), ("xm_cap", i64), ("own_ids", vp), ("own_pos", vp), ("own_count", vp), ("slot_m", vp),
               ("err", vp)])


# name -> (restype, argtypes); every symbol include/orca_hip.h declares
SIGNATURES = {
    "orca_last_error": (ctypes.c_char_p, []),
    "orca_version": (i32, []),
    "orca_csr_create_host": (i32, [i64, vp, vp, vp, vp, i64, i32, ctypes.POINTER(vp)]),
    "orca_csr_create_device": (i32, [i64, vp, vp, vp, vp, i64, vp, ctypes.POINTER(vp)]),
    "orca_csr_from_arrays_host": (i32, [i64, vp, vp, vp, vp, i32, ctypes.POINTER(vp)]),
    "orca_csr_destroy": (i32, [vp]),
    "orca_csr_arrays": (i32, [vp, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(vp), ctypes.POINTER(vp),
                              ctypes.POINTER(vp), ctypes.POINTER(vp)]),
    "orca_csr_copy_host": (i32, [vp, vp, vp, vp, vp]),
    "orca_csr_copy_range_host": (i32, [vp, i64, i64, vp, vp, vp]),
    "orca_sample_recent": (i32, [vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp]),
    "orca_find_before": (i32, [vp, vp, vp, i64, vp, vp]),
    "orca_sample_uniform": (i32, [vp, vp, vp, i64, i32, u64, i64, vp, vp, vp, vp, vp, vp]),
    "orca_time_encode": (i32, [vp, i64, vp, vp, vp, vp]),
    "orca_tgat_blob_floats": (i64, [i32]),
    "orca_tgat_pack": (i32, [i32] + [vp] * 14),
    "orca_tgat_pack_gather": (i32, [i32] + [vp] * 6),
    "orca_tgat_pack_layers": (i32, [i32, i32, vp, vp, vp, i32, vp]),
    "orca_index_positions": (i32, [vp, i64, vp, vp, i64, vp]),
    "orca_scatter_add_flagged": (i32, [vp, vp, vp, i64, vp, vp, i64, vp, vp]),
    "orca_scatter_add_flagged_pair": (i32, [vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, i64, vp, vp]),
    "orca_gru_backward_workspace_floats": (i64, [i32, i64]),
    "orca_gru_backward": (i32, [i32, vp, vp, vp, vp, vp, i64] + [vp] * 11),
    "orca_gru_backward_phase": (i32, [i32, vp, vp, vp, vp, vp, i64] + [vp] * 10 + [i32, vp]),
    "orca_gru_backward_saved": (i32, [i32, vp, vp, vp, vp, vp, i64] + [vp] * 11 + [i32, vp]),
    "orca_gru_update_saving": (i32, [vp, i32, vp, i64, vp, vp, vp, vp, vp, vp]),
    "orca_tgat_backward_workspace_floats": (i64, [i32, i64, i32]),
    "orca_tgat_backward": (i32, [vp, i32, vp, i64, i32] + [vp] * 10),
    "orca_tgat_backward_saved": (i32, [vp, i32, vp, i64, i32] + [vp] * 11),
    "orca_tgat_backward_phase": (i32, [vp, i32, vp, i64, i32] + [vp] * 10 + [i32, vp]),
    "orca_tgat_saved_floats": (i64, [i32, i64, i32]),
    "orca_tgat_forward_saving": (i32, [vp, i32, ctypes.POINTER(TgatInputs), i64, i32, vp, vp, vp, vp]),
    "orca_tgat_dropout_mask": (i32, [u64, i64, i64, i32, ctypes.c_float, vp, vp]),
    "orca_tgat_forward": (i32, [vp, i32, ctypes.POINTER(TgatInputs), i64, i32, vp, vp, vp]),
    "orca_debug_tgat_phase_clock": (i32, [vp]),
    "orca_debug_gru_phase_clock": (i32, [vp]),
    "orca_gru_blob_floats": (i64, [i32]),
    "orca_gru_pack": (i32, [i32, vp, vp, vp, vp, vp, vp]),
    "orca_gru_repack": (i32, [i32, vp, vp, vp, vp, vp, vp]),
    "orca_gru_update": (i32, [vp, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp]),
    "orca_commit_rows": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, vp]),
    "orca_persist_overlay": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, vp]),
    "orca_worklist_build": (i32, [vp, i64, vp, i64, vp, vp, i64, i64, vp, vp, vp]),
    "orca_fill_i32": (i32, [vp, i64, i32, vp]),
    "orca_fill_u8": (i32, [vp, i64, u8, vp]),
    "orca_set_flags": (i32, [vp, vp, vp, i64, vp, u8, i64, vp]),
    "orca_segment_last": (i32, [vp, i64, vp, i64, i64, vp, vp]),
    "orca_segment_last_static": (i32, [vp, i64, i64, vp, vp]),
    "orca_segment_mean": (i32, [vp, i64, i64, vp, vp, vp, i64, i64, vp, vp, vp]),
    "orca_message_store": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, i32, vp, vp, vp, i64, i64, vp, vp, vp, vp]),
    "orca_history_push": (i32, [vp, vp, vp, i64, vp, i64, i64, vp]),
    "orca_gather_rows": (i32, [vp, i64, i64, vp, vp, i64, vp, vp]),
    "orca_scatter_rows": (i32, [vp, i64, i64, vp, i64, vp, vp]),
    "orca_decoder_train_workspace_floats": (i64, [i64]),
    "orca_decoder_train": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "orca_adam_step": (i32, [i32, vp, vp, vp, vp, vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, i64, vp]),
    "orca_find_uncached": (i32, [vp, i64, vp, i64, vp, vp, i64, i64, vp, vp, i64, vp, vp]),
    "orca_engine_create": (i32, [ctypes.POINTER(EngineDesc), ctypes.POINTER(vp)]),
    "orca_engine_destroy": (i32, [vp]),
    "orca_engine_profile": (i32, [vp, i32]),
    "orca_engine_profile_read": (i32, [vp, i32, ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_double)]),
    "orca_engine_set_csr": (i32, [vp, vp]),
    "orca_engine_take_epochs": (i64, [vp, i64]),
    "orca_engine_begin": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp]),
    "orca_engine_layer": (i32, [vp, i32, i64, i64, vp]),
    "orca_engine_train_capture": (i32, [vp, i32, vp, ctypes.c_float, vp, vp]),
    "orca_engine_finish": (i32, [vp, vp, vp, vp, vp, vp]),
    "orca_engine_apply": (i32, [vp, i64, vp]),
    "orca_engine_xm_rows": (i32, [vp, vp]),
    "orca_engine_push_rows": (i32, [vp, vp]),
    "orca_engine_xm_chunk_words": (i64, [i64]),
    "orca_engine_set_sampling": (i32, [vp, i32, u64]),
    "orca_engine_set_overlay_complete": (i32, [vp, i32]),
    "orca_engine_get_overlay_complete": (i32, [vp]),
    "orca_build_roots": (i32, [vp] * 4 + [i64] + [vp] * 5),
    "orca_engine_set_slot_cache": (i32, [vp, vp]),
    "orca_engine_get_slot_cache": (i32, [vp, vp]),
    "orca_slot_cache_materialize": (i32, [vp, i64, vp, vp]),
    "orca_engine_step_budget": (i32, [vp] * 6 + [i64, i32, vp, i64, vp, vp, vp, i64] + [vp] * 9),
    "orca_engine_drain": (i32, [vp, vp]),
    "orca_mrd_plan_host": (i32, [i64, i64, i32, vp, vp, vp, vp, i64, i64, vp, vp, vp, vp]),
    "orca_mrd_plan_device": (i32, [i64, i64, i32, vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp]),
}

_lib = None


class OrcaError(RuntimeError):
  pass


def load():
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise OrcaError(f"{LIB_PATH} is missing: build it with `python -m orca_amd.build` "
                      "(there is no CPU fallback for the HIP path)")
    # torch must be loaded first: its wheel carries its own libamdhip64 and the device pointers we are
    # handed come from that runtime; loading ours first would bring a second HIP runtime into the
    # process (observed: "no ROCm-capable device is detected" from the late-loaded one).
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(L, name)
      fn.restype = res
      fn.argtypes = args
    _lib = L
  return _lib


def check(rc, what=""):
  if rc != 0:
    msg = load().orca_last_error().decode("utf-8", "replace")
    if rc == -4:
      raise IndexError(f"{what}: {msg}")
    if rc == -1:
      raise ValueError(f"{what}: {msg}")
    raise OrcaError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
  """Device (or host) address of a contiguous torch tensor / numpy array; None -> NULL."""
  if t is None:
    return None
  if hasattr(t, "data_ptr"):
    # torch tensors are DEVICE memory at this boundary (host arrays are NumPy arrays): a CPU tensor handed to a kernel
    # is a GPU memory fault on a machine that has a GPU, not an error code
    if not t.is_cuda:
      raise OrcaError("orca_amd: a CPU tensor was passed where device memory is expected (there is no CPU fallback)")
    assert t.is_contiguous(), "orca_amd: tensor must be contiguous"
    return t.data_ptr()
  return t.ctypes.data


_raw_stream = None


def stream():
  """the current HIP stream of the current device, as a raw handle (the fast accessor: the Stream object route costs
  10 us a call, and a training step makes twenty)"""
  global _raw_stream
  import torch
  if _raw_stream is None:
    _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
  if _raw_stream:
    return _raw_stream(torch.cuda.current_device())
  return torch.cuda.current_stream().cuda_stream
