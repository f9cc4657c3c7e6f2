"""ctypes binding of libmpinets_hip.so (the C-ABI declared in include/mpinets_hip.h).

There is no CPU fallback: if the library is missing, fails to load, or a call returns non-zero,
an exception is raised.  PyTorch is used only for device memory and streams -- tensors are passed
as raw device pointers and every call is enqueued on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p
from typing import Optional

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPX_LIB_PATH") or os.path.join(_HERE, "libmpinets_hip.so")  # (override: A/B builds)

P, I, F, L = c_void_p, c_int, c_float, c_int64

# name -> argtypes (the trailing stream argument included); restype is int unless listed below
PROTOTYPES = {
    "mpx_version": [],
    "mpx_device_info": [c_char_p, I, P, P],
    "mpx_prim_frames": [P, P, I, P, P],
    "mpx_cuboid_sdf": [P, P, I, I, P, I, P, P],
    "mpx_cylinder_sdf": [P, P, P, I, I, P, I, P, P],
    "mpx_sphere_sdf": [P, P, I, I, P, I, P, P],
    "mpx_franka_fk": [P, I, F, P, P],
    "mpx_franka_cloud": [P, I, F, P, P, P, I, P, L, I, P],
    "mpx_pose_cloud": [P, I, P, P, I, P, L, I, P],
    "mpx_franka_spheres": [P, I, F, P, P, I, P, P],
    "mpx_franka_collision": [P, I, I, F, P, P, P, I, P, P, I, P, P, P, I, P, P, P],
    "mpx_draw_subset": [I, I, ctypes.c_uint64, I, P, P],
    "mpx_joint_step": [P, P, P, I, P, P, P, P],
    "mpx_franka_success": [P, P, I, F, F, F, P, P, P, P, P],
    "mpx_trajectory_metrics": [P, P, P, P, I, I, F, P, P, P, P, P, P, P],
    "mpx_collision_hinge": [P, L, I, I, I, P, P, I, P, P, P, I, F, P, P, L, I, P],
    "mpx_point_match": [P, P, I, I, F, F, P, P, P],
    "mpx_franka_cloud_grad": [P, I, F, P, P, P, I, P, L, I, P, P],
    "mpx_pack_rows": [P, I, P, I, P, I, I, P, P, P, I, I, I, I, P, P],
    "mpx_pack_rows_grad": [P, I, P, P, P, I, I, I, I, P, I, P],
    "mpx_pack_rows_ld": [P, I, P, I, P, I, I, P, P, P, I, I, I, I, P, I, P],
    "mpx_pack_rows_grad_ld": [P, I, I, P, P, P, I, I, I, I, P, I, P],
    "mpx_segment_max": [P, I, P, L, P, I, P, P],
    "mpx_segment_max_grad": [P, I, P, L, I, P, P],
    "mpx_segment_max_grad_act": [P, I, P, P, I, P, L, I, I, P, P],
    "mpx_linear_segmax": [P, I, P, P, I, I, I, I, P, L, P, P, I, P, P],
    "mpx_linear_segmax_bf16x3": [P, I, P, P, I, I, I, I, P, L, P, P, I, P, P],
    "mpx_pool_wgrad_scratch": [L, I, I],
    "mpx_pool_wgrad": [P, I, P, P, I, L, I, I, P, I, I, P, P, P, P],
    "mpx_pool_dgrad": [P, I, P, P, I, P, L, I, I, P, I, P, I, I, I, I, P, I, P],
    "mpx_batch_configs": [P, L, I, P, P, P, F, ctypes.c_uint64, L, I, I, F, P, P, P, P, P, P],
    "mpx_gather_rows": [P, P, I, I, P, P],
    "mpx_depth_render": [P, F, F, F, F, I, I, I, P, P, I, P, P, P, I, P, P, I, F, P, P],
    "mpx_depth_select": [P, P, F, F, F, F, I, I, I, I, ctypes.c_uint64, L, P, L, I, P, P],
    "mpx_scene_cloud": [P, P, P, I, P, P, P, P, I, I, I, ctypes.c_uint64, L, P, P, P, P, L, I, I, P],
    "mpx_set_variant": [I, I],
    "mpx_get_variant": [I],
    "mpx_fps": [P, I, I, I, I, P, P, I, P],
    "mpx_ball_query": [P, I, P, I, I, I, I, F, I, P, P, P],
    "mpx_ball_query_hits": [P, I, P, I, I, I, I, F, I, P, P, P],
    "mpx_sort_queries": [P, L, I, P, P, P],
    "mpx_group_points": [P, I, P, I, P, I, I, P, I, I, I, I, P, P],
    "mpx_sa_mlp": [P, I, P, I, P, I, I, P, P, I, I, I, I, P, I, I, I, P, I, I, P],
    "mpx_sa_mlp_factored": [P, P, P, P, I, I, I, I, P, I, I, I, I, P, I, P],
    "mpx_sa_mlp_bf16x3_factored_wants_order": [],
    "mpx_sa_mlp_bf16x3_factored": [P, P, P, P, P, I, I, I, I, P, I, I, I, I, P, I, P],
    "mpx_sa_pack_size": [I, I, I, I],
    "mpx_sa_pack_weights": [P, P, P, P, P, P, I, I, I, I, P, P],
    "mpx_sa_mlp_bf16x3": [P, I, P, I, P, I, I, P, P, P, I, I, I, I, P, I, I, I, P, I, I, P],
    "mpx_sa_mlp_bf16x3_wants_order": [I, I, I, I, I],
    "mpx_sa_pack_bf16x3_size": [I, I, I, I],
    "mpx_sa_pack_bf16x3": [P, P, P, P, P, P, I, I, I, I, P, P],
    "mpx_sa3_pack_size": [I, I, I, I],
    "mpx_sa3_pack_weights": [P, I, P, P, P, P, P, I, I, I, I, P, P],
    "mpx_sa3_chain": [P, I, I, I, P, I, I, I, I, P, I, P],
    "mpx_sa3_chain_probe": [P, I, I, P, P, I, P, P],
    "mpx_sa2_bf16x3_set_probe": [P],
    "mpx_linear": [P, I, P, P, I, I, I, I, P, I, P],
    "mpx_linear_workspace": [I, I, I],
    "mpx_linear_ws": [P, I, P, P, I, I, I, I, P, I, P, L, P],
    "mpx_linear_rowmax": [P, I, P, P, I, I, I, I, P, I, P],
    "mpx_split_bf16": [P, I, L, I, P, I, P],
    "mpx_linear_bf16x3": [P, I, P, P, I, I, I, I, P, I, P],
    "mpx_linear_rowmax_bf16x3": [P, I, P, P, I, I, I, I, P, I, P],
    "mpx_linear_bf16x3_to_pairs": [P, I, P, P, I, I, I, I, P, I, P],
    "mpx_linear_bf16x3_pairs": [P, I, P, P, I, I, I, I, P, I, P, I, P],
    "mpx_linear_rowmax_bf16x3_pairs": [P, I, P, P, I, I, I, I, P, I, P, I, P],
    "mpx_sa3_front_bf16x3_pack_size": [I, I, I],
    "mpx_sa3_front_bf16x3_pack": [P, I, P, P, P, I, I, I, P, P],
    "mpx_sa3_front_bf16x3_w3_pairs": [P, I, I, P, P],
    "mpx_sa3_front_bf16x3": [P, I, I, I, P, P, I, P],
    "mpx_sa3_front_bf16x3_probe": [P, I, I, P, P, I, P, P],
    "mpx_groupnorm_leaky_grad": [P, P, P, P, I, I, I, F, P, P, P, P, P],
    "mpx_act_backward": [P, P, L, I, P, P],
    "mpx_linear_dact": [P, I, P, I, I, I, P, I, I, P, I, P],
    "mpx_linear_bf16x3_dact": [P, I, P, I, I, I, P, I, I, P, I, P],
    "mpx_linear_wgrad_bf16x3": [P, I, P, I, I, I, I, P, P, P, P],
    "mpx_linear_wgrad_scratch": [I, I, I],
    "mpx_linear_wgrad": [P, I, P, I, I, I, I, P, P, P, P],
    "mpx_groupnorm_leaky": [P, P, P, I, I, I, F, P, P],
    "mpx_groupnorm_leaky_to_pairs": [P, P, P, I, I, I, F, P, I, P],
    "mpx_rowmax": [P, I, I, I, I, P, I, P],
    "mpx_append_columns": [P, I, I, I, L, P, I, I, P],
    "mpx_policy_workspace": [I, I],
    "mpx_policy_forward": [P, P, I, P, I, P, P, L, P],
    "mpx_rollout_workspace": [I, I],
    "mpx_rollout_step": [P, P, P, I, P, P, I, P, P, P, L, P],
    "mpx_rollout": [P, P, P, P, I, P, P, I, P, P, P, L, P],
}
RESTYPES = {"mpx_last_error": c_char_p, "mpx_sa_pack_size": c_int64, "mpx_sa3_front_bf16x3_pack_size": c_int64, "mpx_sa_pack_bf16x3_size": c_int64,
            "mpx_linear_wgrad_scratch": c_int64, "mpx_pool_wgrad_scratch": c_int64, "mpx_linear_workspace": c_int64, "mpx_policy_workspace": c_int64, "mpx_rollout_workspace": c_int64}

_lib: Optional[ctypes.CDLL] = None


class MpxError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library (raises if it has not been built: run ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MpxError(
            f"{LIB_PATH} not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C motion-policy-networks_amd/csrc`). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.mpx_last_error.restype = c_char_p
    lib.mpx_last_error.argtypes = []
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, c_int)
    _lib = lib
    return lib


def exported_symbols():
    """Names the header declares (used by the CPU-side symbol test)."""
    return ["mpx_last_error"] + list(PROTOTYPES)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """The current torch stream of the current device as a hipStream_t.  (The raw getter returns the handle without
    building a ``torch.cuda.Stream`` object: the public accessor was ~70 % of a call's host time, and a training step at the
    reference's batch of 10 is bound by the host.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return t.data_ptr()


# name -> list of (start_event, end_event) recorded around each launch of that entry point while
# profiling is on (bench.py's live per-kernel timing; events sit on the launch stream).
PROFILE: Optional[dict] = None
# entry points timed under another one's name (same kernels, another output contract)
PROFILE_KEY = {"mpx_ball_query_hits": "mpx_ball_query"}


def profile_start(*names: str) -> None:
    global PROFILE
    PROFILE = {n: [] for n in names}


def profile_stop() -> dict:
    """-> {name: [milliseconds per launch, in call order]} (synchronises the device)."""
    global PROFILE
    prof, PROFILE = PROFILE or {}, None
    torch.cuda.synchronize()
    return {n: [a.elapsed_time(b) for a, b in evs] for n, evs in prof.items()}


def call(name: str, *args):
    """Call ``name`` with the current torch stream appended; raise on a non-zero status."""
    lib = _lib or load()
    fn = getattr(lib, name)
    if PROFILE is None:  # (the common case first: one attribute lookup, one stream query, the call)
        rc = fn(*args, stream_ptr())
    else:
        evs = None
        key = PROFILE_KEY.get(name, name)
        if key in PROFILE:
            evs = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            evs[0].record()
        rc = fn(*args, stream_ptr())
        if evs is not None:
            evs[1].record()
            PROFILE[key].append(evs)
    if rc != 0:
        raise MpxError(f"{name} failed ({rc}): {lib.mpx_last_error().decode()}")


def require_cuda(*tensors: Optional[torch.Tensor]):
    """No CPU fallback -- and no silent cross-device launches: every call is enqueued on torch's current stream of
    the CURRENT device, so operands that live on another GPU (``model.to('cuda:1')`` without
    ``torch.cuda.set_device(1)``) are rejected instead of being launched on device 0's stream."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise MpxError("CPU tensors are not supported by the HIP engine (no CPU fallback)")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise MpxError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: call "
                           f"torch.cuda.set_device({t.device.index}) (or use `with torch.cuda.device(...)`) first")


def f32c(t: torch.Tensor) -> torch.Tensor:
    """float32 + contiguous (no copy when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def i32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.int32:
        t = t.int()
    return t if t.is_contiguous() else t.contiguous()
