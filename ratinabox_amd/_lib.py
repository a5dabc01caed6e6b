"""ctypes binding of libriab_hip.so (include/riab_hip.h).

The library is the product's only compute path.  If it is missing it is built
with hipcc; if that fails, importing this module RAISES — there is no CPU
fallback (the CPU restatement under oracle/ is test infrastructure only)."""
import ctypes as C
import os

# torch must be imported BEFORE libriab_hip.so is loaded: torch bundles its own HIP runtime
# (soname libamdhip64.so.7, requested by torch as "libamdhip64.so"); loading ours first would
# put a second runtime in the process and the later one finds no device.
import torch  # noqa: F401

from . import _build

ABI_VERSION = 8
MAX_WALLS = 64
STATE_ROWS = 12
HIST_ROWS = 8

# row indices (mirror the enums in riab_hip.h)
S_POS_X, S_POS_Y, S_VEL_X, S_VEL_Y, S_ROT_VEL, S_MVEL_X, S_MVEL_Y, S_MROT_VEL, S_HD_X, S_HD_Y, S_DIST, S_DWALL = range(12)
H_POS_X, H_POS_Y, H_VEL_X, H_VEL_Y, H_HD_X, H_HD_Y, H_ROT_VEL, H_DIST = range(8)

PC_DESCRIPTIONS = {"gaussian": 0, "gaussian_threshold": 1, "diff_of_gaussians": 2, "one_hot": 3, "top_hat": 4}
GEOMETRIES = {"euclidean": 0, "line_of_sight": 1, "geodesic": 2}
GC_DESCRIPTIONS = {"rectified_cosines": 0, "shifted_cosines": 1}


class RiabEnv(C.Structure):
    _fields_ = [("extent", C.c_double * 4), ("scale", C.c_double), ("periodic", C.c_int32),
                ("n_walls", C.c_int32), ("walls", C.c_void_p), ("polygon", C.c_int32), ("n_boundary", C.c_int32),
                ("hole_mask", C.c_uint64)]


class RiabMotion(C.Structure):
    _fields_ = [("dt", C.c_double), ("rot_theta_kw", C.c_double), ("rot_sigma_kw", C.c_double),
                ("rot_drift_kw", C.c_double), ("speed_theta_kw", C.c_double), ("speed_sigma_kw", C.c_double),
                ("speed_mean_kw", C.c_double), ("speed_mean", C.c_double), ("speed_std_is_zero", C.c_int32),
                ("has_drift", C.c_int32), ("drift_theta", C.c_double), ("wall_repel_strength_kw", C.c_double),
                ("wall_repel_distance_kw", C.c_double), ("thigmotaxis_kw", C.c_double), ("hd_tau", C.c_double),
                ("wall_grid", C.c_void_p), ("wall_grid_n", C.c_int32), ("wall_grid_wd", C.c_double),
                ("wall_grid_lmax", C.c_double)]


class RiabRateIO(C.Structure):
    _fields_ = [("pos_x", C.c_void_p), ("pos_y", C.c_void_p), ("hd_x", C.c_void_p), ("hd_y", C.c_void_p),
                ("pos_ld", C.c_int64), ("T", C.c_int64), ("B", C.c_int64), ("rates", C.c_void_p),
                ("spikes", C.c_void_p), ("u_in", C.c_void_p), ("dt", C.c_float), ("min_fr", C.c_float),
                ("max_fr", C.c_float), ("seed", C.c_uint64), ("step0", C.c_uint64), ("agent_id0", C.c_int64),
                ("pop_id", C.c_int32)]


class RiabFFInput(C.Structure):
    _fields_ = [("rates", C.c_void_p), ("wt", C.c_void_p), ("n_in", C.c_int32)]


class RiabPopulation(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("io", RiabRateIO), ("rates_base", C.c_void_p),
                ("spikes_base", C.c_void_p), ("capacity_rows", C.c_int64), ("table", C.c_void_p),
                ("description", C.c_int32), ("geometry", C.c_int32), ("top_hat_width", C.c_float), ("f0", C.c_float),
                ("test_dirs", C.c_void_p), ("ray_rden", C.c_void_p), ("K", C.c_int32), ("egocentric", C.c_int32),
                ("vm_table", C.c_void_p), ("inv_norm", C.c_void_p), ("cell_rows", C.c_void_p), ("windows", C.c_void_p),
                ("bvc_xch", C.c_void_p), ("bvc_xch_count", C.c_void_p),
                ("objects", C.c_void_p),
                ("object_types", C.c_void_p), ("n_objects", C.c_int32), ("walls_occlude", C.c_int32),
                ("one_sigma_speed", C.c_float), ("targets", C.c_void_p), ("n_anchors", C.c_int32),
                ("noise_state", C.c_void_p), ("noise_theta_dt", C.c_float), ("noise_sigma_dt", C.c_float),
                ("n_inputs", C.c_int32), ("input_index", C.c_int32 * 8), ("input_wt", C.c_void_p * 8),
                ("bias", C.c_void_p), ("activation", C.c_int32), ("act_params", C.c_float * 4),
                ("rates_prime", C.c_void_p)]


class RiabSimulate(C.Structure):
    _fields_ = [("env", C.POINTER(RiabEnv)), ("motion", C.POINTER(RiabMotion)), ("state", C.c_void_p), ("B", C.c_int64),
                ("agent_id0", C.c_int64), ("drift", C.c_void_p), ("noise", C.c_void_p), ("forced_pos", C.c_void_p),
                ("resample_pos", C.c_void_p), ("seed", C.c_uint64), ("step0", C.c_uint64), ("T", C.c_int32),
                ("n_pops", C.c_int32), ("pops", C.POINTER(RiabPopulation)), ("hist", C.c_void_p), ("diag", C.c_void_p),
                ("ctrl", C.c_void_p), ("timed_pop", C.c_int32), ("timing_mode", C.c_int32),
                ("watch", C.c_void_p), ("n_watch", C.c_int32)]


class RiabWatch(C.Structure):
    """A host array a cached device table was built from + the snapshot taken then (riab_simulate compares them)."""
    _fields_ = [("live", C.c_void_p), ("snapshot", C.c_void_p), ("bytes", C.c_int64)]


POP_SIZE = C.sizeof(RiabPopulation)


class RiabTask(C.Structure):
    _fields_ = [("goals", C.c_void_p), ("n_pool", C.c_int32), ("goalorder", C.c_int32), ("terminate_delay", C.c_double),
                ("pad_reward", C.c_double * 5), ("default_reward_level", C.c_double)]


# rows of the per-lane task state tensor (include/riab_hip.h RIAB_TS_*)
TS_N_GOALS, TS_DELAYED, TS_PAD_START, TS_N_REWARDS, TS_EPISODE, TS_EP_START, TS_EP_ANY_ENDED = range(7)
TS_STEPS_ACTIVE, TS_STEPS_INACTIVE, TS_R_MAX, TS_R_MIN, TS_STARTED = 7, 8, 9, 10, 11
TASK_MAX_GOALS, TASK_MAX_REWARDS, TASK_MAX_POOL = 16, 32, 64
TS_GOAL_LIST = 12
TS_RW_STATE = TS_GOAL_LIST + TASK_MAX_GOALS
TS_RW_EXPIRE = TS_RW_STATE + TASK_MAX_REWARDS
TS_RW_SRC = TS_RW_EXPIRE + TASK_MAX_REWARDS
TS_ROWS = TS_RW_SRC + TASK_MAX_REWARDS
GOAL_TIME_ELAPSED = -2
DECAYS = {"constant": 0, "linear": 1, "exponential": 2, "none": 3}
GOALORDERS = {"nonsequential": 0, "sequential": 1}
TD_REWARD_OVERFLOW, TD_LATE_COMPLETIONS, TD_EPLOG_OVERFLOW, TD_RESETS = range(4)
# rows of the shared state of a task whose lanes are the agents of one world (RIAB_TW_*)
TW_N_GOALS, TW_DELAYED, TW_PAD_START, TW_EPISODE, TW_EP_START, TW_EP_ANY_ENDED, TW_STARTED = range(7)
TW_TERMINAL, TW_GOAL_LIST, TW_ROWS = 7, 8, 24

POP_KINDS = {"place": 0, "grid": 1, "hdc": 2, "bvc": 3, "ovc": 4, "ff": 5, "velocity": 6, "speed": 7, "random_spatial": 8}
EINVAL = -1
EALIGN = -2
EFULL = -5
EUNSUPPORTED = -4
EPARTIAL = -6
ECHANGED = -7
STREAMER_OPT_GATE, STREAMER_OPT_POLL_MAX, STREAMER_OPT_HEAD_ROWS = 0, 1, 2
STREAMER_OPT_SIDE_STREAM, STREAMER_OPT_STEP_NS, STREAMER_OPT_LEAD_MBPS = 3, 4, 5
STREAMER_OPT_STRICT, STREAMER_OPT_SPIN_LIMIT = 6, 7
GATE_ALWAYS, GATE_WHEN_BUSY, GATE_RESERVED = 0, 1, 2
CTRL_STARTED, CTRL_TIMEOUTS, CTRL_ABORT, CTRL_SERIALISED, CTRL_STAMPS, CTRL_TRAJ_STAMPS, CTRL_PROGRESS = 0, 1, 2, 3, 8, 12, 32  # riab_hip.h RIAB_CTRL_*


STEP1_SYNC_STRIDE, STEP1_SYNC_TAIL, STEP1_SYNC_TIMEOUTS = 64, 16, 0   # riab_hip.h RIAB_STEP1_SYNC_*
STEP1_SYNC_FIRST_BAD, STEP1_SYNC_LAST_BAD, STEP1_SYNC_FATAL = 1, 2, 3
STEP1_MAX_POPS = 4                                                      # csrc/riab_device.h RIAB_STEP1_MAX_POPS
WALL_GRID_MAX = 16                                                      # riab_hip.h RIAB_WALL_GRID_MAX
CU_PROBE_WORDS = 4097                                                   # riab_hip.h RIAB_CU_PROBE_WORDS
STEP1_MAIL_STRIDE = 1088                                                # riab_hip.h RIAB_STEP1_MAIL_STRIDE


def step1_sync_tail(B):
    """Index of the counters behind the arrival words."""
    return ((int(B) + 255) // 256) * STEP1_SYNC_STRIDE


def step1_sync_words(B):
    """RIAB_STEP1_SYNC_WORDS(B): arrival words, counters and the prepared wall table, for B agents."""
    segs = (int(B) + 255) // 256
    return segs * STEP1_SYNC_STRIDE + STEP1_SYNC_TAIL + 12 * MAX_WALLS + 4 + segs * STEP1_MAIL_STRIDE


def ctrl_words(B):
    """RIAB_CTRL_WORDS(B): control words of the flag-coupled pipeline for B agents."""
    return CTRL_PROGRESS + 32 * ((int(B) + 255) // 256)

ACTIVATIONS = {"linear": 0, "sigmoid": 1, "relu": 2, "tanh": 3, "retanh": 4, "softmax": 5}

# name -> (restype, argtypes): every symbol include/riab_hip.h declares
PROTOTYPES = {
    "riab_agent_step": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabMotion), C.c_void_p, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "riab_place_cells": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabRateIO), C.c_void_p, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_float, C.c_void_p]),
    "riab_grid_cells": (C.c_int, [C.POINTER(RiabRateIO), C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "riab_head_direction_cells": (C.c_int, [C.POINTER(RiabRateIO), C.c_void_p, C.c_int32, C.c_void_p]),
    "riab_env_pairwise": (C.c_int, [C.POINTER(RiabEnv), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "riab_env_vectors_from_walls": (C.c_int, [C.POINTER(RiabEnv), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                              C.c_void_p]),
    "riab_env_check_wall_collisions": (C.c_int, [C.POINTER(RiabEnv), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_int64, C.c_void_p, C.c_void_p]),
    "riab_env_boundary_conditions": (C.c_int, [C.POINTER(RiabEnv), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_int32, C.c_void_p]),
    "riab_random_spatial_neurons": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabRateIO), C.c_void_p, C.c_int32,
                                              C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "riab_agent_vector_cells": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabRateIO), C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "riab_velocity_cells": (C.c_int, [C.POINTER(RiabRateIO), C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "riab_speed_cell": (C.c_int, [C.POINTER(RiabRateIO), C.c_float, C.c_void_p]),
    "riab_boundary_vector_cells": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabRateIO), C.c_void_p, C.c_void_p, C.c_int32,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                             C.c_void_p]),
    "riab_boundary_vector_cells_windowed": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabRateIO), C.c_void_p, C.c_void_p,
                                                      C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "riab_object_vector_cells": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabRateIO), C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "riab_spikes": (C.c_int, [C.POINTER(RiabRateIO), C.c_int32, C.c_void_p]),
    "riab_neuron_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float,
                                    C.c_uint64, C.c_uint64, C.c_int32, C.c_int64, C.c_void_p]),
    "riab_feedforward": (C.c_int, [C.POINTER(RiabFFInput), C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int64,
                                   C.c_int32, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p]),
    "riab_fill": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "riab_plan_create": (C.c_void_p, [C.POINTER(RiabEnv), C.POINTER(RiabMotion), C.c_void_p, C.c_int64, C.c_int64,
                                      C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "riab_plan_destroy": (None, [C.c_void_p]),
    "riab_plan_set_motion": (C.c_int, [C.c_void_p, C.POINTER(RiabMotion), C.c_void_p]),
    "riab_plan_set_forced": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "riab_plan_set_agent_history": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "riab_plan_add": (C.c_int, [C.c_void_p, C.POINTER(RiabPopulation)]),
    "riab_plan_set_population_history": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    "riab_plan_rows_free": (C.c_int64, [C.c_void_p]),
    "riab_plan_step_index": (C.c_uint64, [C.c_void_p]),
    "riab_plan_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "riab_plan_step_agent": (C.c_int, [C.c_void_p, C.c_void_p]),
    "riab_plan_step_population": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "riab_plan_set_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "riab_plan_discard_ahead": (C.c_int, [C.c_void_p]),
    "riab_plan_set_compute_units": (C.c_int, [C.c_void_p, C.c_int32]),
    "riab_probe_compute_units": (C.c_int, [C.c_void_p, C.c_void_p]),
    "riab_plan_info": (C.c_int64, [C.c_void_p, C.c_int32]),
    "riab_task_step": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabTask), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "riab_task_goal_vector": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabTask), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "riab_plan_set_task": (C.c_int, [C.c_void_p, C.POINTER(RiabTask), C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64,
                                     C.c_uint64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_double]),
    "riab_plan_task_clock": (C.c_double, [C.c_void_p]),
    "riab_task_reset": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabTask), C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                  C.c_double, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "riab_task_world_step": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabTask), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "riab_task_world_reset": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabTask), C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                        C.c_double, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "riab_plan_set_task_world": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "riab_task_world_goal_vector": (C.c_int, [C.POINTER(RiabEnv), C.POINTER(RiabTask), C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "riab_streamer_create": (C.c_void_p, []),
    "riab_streamer_destroy": (None, [C.c_void_p]),
    "riab_simulate": (C.c_int, [C.c_void_p, C.POINTER(RiabSimulate), C.c_void_p]),
    "riab_streamer_warmup": (C.c_int, [C.c_void_p, C.c_void_p]),
    "riab_watch_compare": (C.c_int, [C.c_void_p, C.c_int32]),
    "riab_streamer_configure": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "riab_streamer_last_rate_ms": (C.c_float, [C.c_void_p]),
    "riab_streamer_last_form": (C.c_int, [C.c_void_p]),
    "riab_streamer_info": (C.c_int64, [C.c_void_p, C.c_int32]),
    "riab_host_wait_spin": (C.c_int, [C.c_int32]),
    "riab_set_option": (C.c_int, [C.c_int32, C.c_int32]),
    "riab_abi_sizeof": (C.c_int64, [C.c_int32]),
    "riab_abi_version": (C.c_int, []),
    "riab_strerror": (C.c_char_p, [C.c_int]),
}


class RiabError(RuntimeError):
    pass


def _load():
    path = _build.LIB_PATH
    override = os.environ.get("RIAB_HIP_LIB")  # kernel experiments: load an alternative build
    if override:
        path = override
    elif _build.is_stale():
        try:
            path = _build.build()
        except Exception as e:  # noqa: BLE001
            if not os.path.exists(path):
                raise ImportError(
                    "ratinabox_amd: libriab_hip.so is missing and could not be built with hipcc "
                    f"({e}). There is no CPU fallback; build it with `python -m ratinabox_amd._build`.") from e
            # an older library is there but the sources are newer and cannot be rebuilt here: say so (the struct
            # checks below still refuse a library whose layouts differ from this binding's)
            import warnings
            warnings.warn(f"ratinabox_amd: libriab_hip.so is older than its sources and could not be rebuilt ({e}); "
                          "loading the existing library", RuntimeWarning)
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.riab_abi_version() != ABI_VERSION:
        raise ImportError(f"libriab_hip.so ABI {lib.riab_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    # the ctypes mirrors must have the layouts the library was compiled with (a stale library with the same
    # version number would otherwise corrupt memory silently)
    mirrors = ((0, RiabEnv), (1, RiabMotion), (2, RiabRateIO), (3, RiabPopulation), (4, RiabTask), (5, RiabFFInput),
               (7, RiabSimulate), (8, RiabWatch))
    for which, cls in mirrors:
        if lib.riab_abi_sizeof(which) != C.sizeof(cls):
            raise ImportError(f"libriab_hip.so: sizeof({cls.__name__}) is {lib.riab_abi_sizeof(which)} in the library, "
                              f"{C.sizeof(cls)} in the binding: rebuild (python -m ratinabox_amd._build --force)")
    if lib.riab_abi_sizeof(6) != TS_ROWS:
        raise ImportError(f"libriab_hip.so: RIAB_TS_ROWS is {lib.riab_abi_sizeof(6)}, the binding has {TS_ROWS}: rebuild")
    return lib, path


lib, LIB_PATH = _load()


OPTIONS = {"traj_kernel": 0, "fused_task": 1, "bvc_box": 2, "nt_stores": 3, "pub_single_rows": 4, "poll_sleep": 5,
           "fused_step": 6, "step1_spin": 7, "step1_residency": 8}   # riab_hip.h RIAB_OPT_*


def set_option(name, value):
    """riab_set_option: an A/B switch of the library (tests, comparisons); returns the previous value."""
    old = lib.riab_set_option(OPTIONS[name], int(value))
    if old < 0:
        raise RiabError(f"riab_set_option({name}, {value}) refused")
    return old


# environment variables set BEFORE the import select the same switches for a whole process (tools, A/B runs)
for _name, _opt, _val in (("RIAB_NO_PC", "traj_kernel", 1), ("RIAB_TRAJ2", "traj_kernel", 2), ("RIAB_NO_FUSED_TASK", "fused_task", 0),
                          ("RIAB_NO_BVC_BOX", "bvc_box", 0), ("RIAB_NT_STORES_WIDE", "nt_stores", 1),
                          ("RIAB_NO_FUSED_STEP", "fused_step", 0), ("RIAB_FUSED_STEP_PLAIN_STORES", "fused_step", 2)):
    if os.environ.get(_name):
        set_option(_opt, _val)
for _name, _opt in (("RIAB_PUB_SINGLE_ROWS", "pub_single_rows"), ("RIAB_POLL_SLEEP", "poll_sleep"),
                    ("RIAB_STEP1_SPIN", "step1_spin"), ("RIAB_STEP1_RESIDENCY", "step1_residency")):
    if os.environ.get(_name):
        set_option(_opt, int(os.environ[_name]))


def strerror(code):
    return lib.riab_strerror(int(code)).decode()


def check(code, what):
    """Raise on a non-zero return code of an ABI call."""
    if code != 0:
        raise RiabError(f"{what} failed with code {code}: {strerror(code)}")


_ENV_DATA = getattr(os.environ, "_data", None)


def env(name, default=None):
    """os.environ.get for the A/B switches read on latency-critical paths: `os.environ.get` encodes the key and
    goes through two wrappers (3-4 us per lookup); CPython keeps the raw mapping in `os.environ._data`."""
    if _ENV_DATA is not None:
        v = _ENV_DATA.get(name.encode() if os.name != "nt" else name.upper())
        return default if v is None else (v.decode() if isinstance(v, bytes) else v)
    return os.environ.get(name, default)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    """hipStream_t of torch's current stream on the current device (the raw handle straight from the C
    API: torch.cuda.current_stream() costs ~8 us of Python per call, this ~0.5 us)."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_COMPUTE_UNITS = {}


def compute_units(device_index):
    """The compute units this process's workgroups on torch's current stream of the device really land on
    (riab_probe_compute_units), measured once per device: a process under HSA_CU_MASK / ROC_GLOBAL_CU_MASK, or on a
    partition of the chip, is told the whole device by hipGetDeviceProperties.  `RIAB_COMPUTE_UNITS=n` skips the probe."""
    n = _COMPUTE_UNITS.get(device_index)
    if n is None:
        import torch
        forced = os.environ.get("RIAB_COMPUTE_UNITS")
        if forced:
            n = int(forced)
        else:
            with torch.cuda.device(device_index):
                scratch = torch.zeros(CU_PROBE_WORDS, dtype=torch.int32, device=f"cuda:{device_index}")
                check(lib.riab_probe_compute_units(ptr(scratch), current_stream()), "riab_probe_compute_units")
                n = int(scratch[CU_PROBE_WORDS - 1].item())
            props = torch.cuda.get_device_properties(device_index).multi_processor_count
            if n < 1 or n > props:   # (a probe that makes no sense: what the runtime says)
                n = props
        _COMPUTE_UNITS[device_index] = n
    return n
