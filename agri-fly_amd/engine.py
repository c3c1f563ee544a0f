"""ctypes binding of include/agrifly_engine.h (the C ABI of the HIP engine).

Mirrors the C entry points one to one; see the header for which reference
interface (agri-fly file:line) each call replaces.  Host arrays are numpy,
planar ``[components, count]``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# AGRIFLY_ENGINE_LIB selects another build of the same library (kernel A/B runs)
_LIB = os.environ.get("AGRIFLY_ENGINE_LIB") or os.path.join(_HERE, "lib", "libagrifly_engine.so")

AFE_F32, AFE_F64 = 0, 1
AFE_SEED_REFERENCE, AFE_SEED_DECORRELATED, AFE_SEED_COUNTER = 0, 1, 2
AFE_STEP_LAUNCH, AFE_STEP_PERSISTENT, AFE_STEP_AUTO, AFE_STEP_RESIDENT = 0, 1, 2, 3

# every symbol include/agrifly_engine.h declares (checked by tests/test_abi.py)
ABI_FUNCTIONS = [
    "afe_params_from_type", "afe_type_from_id", "afe_create", "afe_create_host_visible", "afe_destroy",
    "afe_last_error", "afe_status_string", "afe_abi_version", "afe_set_stream",
    "afe_set_type_table", "afe_set_vehicle_types", "afe_set_logic_period",
    "afe_set_imu_noise", "afe_set_state", "afe_get_state", "afe_set_state_f32",
    "afe_get_state_f32", "afe_set_rng_state", "afe_get_rng_state",
    "afe_set_motor_cmds", "afe_set_external_force", "afe_set_external_torque",
    "afe_step", "afe_steps_until_tick", "afe_sync", "afe_time_us",
    "afe_logic_ticks", "afe_get_imu", "afe_plan_ticks", "afe_get_device_view",
    "afe_algorithmic_bytes_per_step", "afe_event_create", "afe_event_destroy",
    "afe_event_record", "afe_event_elapsed_ms", "afe_pack_positions",
    "afe_nearest_neighbour", "afe_selftest_normals", "afe_selftest_normals_f32",
    "afe_rates_logic_params_from_type", "afe_set_rates_logic", "afe_set_rates_commands",
    "afe_get_motor_cmds", "afe_checkpoint_size", "afe_save_checkpoint", "afe_load_checkpoint",
    "afe_radio_create_rates_command", "afe_radio_create_position_command",
    "afe_radio_create_acceleration_command", "afe_radio_create_simple_command", "afe_radio_decode",
    "afe_telemetry_encode", "afe_telemetry_decode", "afe_set_commands_from_radio",
    "afe_set_max_fused_steps", "afe_set_addressing", "afe_step_kernel_info", "afe_planner_default_config", "afe_planner_samples", "afe_rappids_plan",
    "afe_set_split_stepping", "afe_rappids_plan_device", "afe_planner_release_scratch", "afe_camera_default", "afe_camera_default_mount", "afe_scene_create",
    "afe_scene_destroy", "afe_scene_info", "afe_scene_set_walk", "afe_render_depth", "afe_render_depth_engine", "afe_render_depth_stats",
    "afe_device_alloc", "afe_device_free", "afe_device_download", "afe_scene_check_hierarchy",
    "afe_comm_unique_id", "afe_comm_create", "afe_comm_info", "afe_comm_destroy", "afe_comm_last_error",
    "afe_gather_positions", "afe_group_create", "afe_group_destroy", "afe_group_size", "afe_group_shard",
    "afe_group_step", "afe_group_sync", "afe_group_gather_positions", "afe_group_last_error", "afe_group_peer_access",
    "afe_nearest_neighbour_grid", "afe_neighbour_grid_info", "afe_set_neighbour_grid_refresh", "afe_set_neighbour_sort_reuse", "afe_nearest_neighbour_bruteforce",
    "afe_uwb_create", "afe_uwb_destroy", "afe_uwb_set_noise", "afe_uwb_draw", "afe_uwb_range",
    "afe_set_step_mode", "afe_steps_completed", "afe_persistent_running", "afe_stream_probe",
    "afe_set_noise_seed", "afe_set_gust_process", "afe_get_external_force", "afe_nearest_neighbour_async", "afe_query_sync",
    "afe_gather_exchange", "afe_set_cache_policy", "afe_grid_time", "afe_cache_policy_in_use", "afe_set_resident_queue",
    "afe_has_dev_hooks", "afe_persistent_kernarg_layout", "afe_group_set_staged_copies",
]


class AfeError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("agrifly_engine status %d: %s" % (status, message))
        self.status = status


class VehicleParams(C.Structure):
    """afe_vehicle_params == the Quadcopter_T constructor arguments
    (reference Components/Components/Simulation/Quadcopter_T.hpp:24-32)."""
    _fields_ = [
        ("mass", C.c_double),
        ("inertia", C.c_double * 9),
        ("arm_length", C.c_double),
        ("com_error", C.c_double * 3),
        ("motor_min_speed", C.c_double),
        ("motor_max_speed", C.c_double),
        ("prop_thrust_from_speed_sqr", C.c_double),
        ("prop_torque_from_speed_sqr", C.c_double),
        ("motor_time_const", C.c_double),
        ("motor_inertia", C.c_double),
        ("lin_drag_coeff_b", C.c_double * 3),
        ("imu_yaw", C.c_float),
        ("imu_pitch", C.c_float),
        ("imu_roll", C.c_float),
    ]

    def copy(self):
        other = VehicleParams()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(VehicleParams))
        return other

    @property
    def hover_speed(self):
        """per-motor speed at which 4 k_f w^2 = m g"""
        return float(np.sqrt(self.mass * 9.81 / (4.0 * self.prop_thrust_from_speed_sqr)))


class RatesLogicParams(C.Structure):
    """afe_rates_logic_params: the rates-control slice of Onboard::QuadcopterLogic."""
    _fields_ = [
        ("mass", C.c_float), ("inertia", C.c_float * 9),
        ("ang_vel_time_const_xy", C.c_float), ("ang_vel_time_const_z", C.c_float),
        ("arm_length", C.c_float), ("prop_thrust_from_speed_sqr", C.c_float),
        ("prop_torque_from_thrust", C.c_float), ("prop0_spin_dir", C.c_int),
        ("max_thrust_per_propeller", C.c_float), ("min_thrust_per_propeller", C.c_float),
        ("max_cmd_total_thrust", C.c_float),
        ("imu_yaw", C.c_float), ("imu_pitch", C.c_float), ("imu_roll", C.c_float),
        ("gyro_lowpass_cutoff", C.c_float),
    ]


class RadioMessage(C.Structure):
    _fields_ = [("type", C.c_uint8), ("flags", C.c_uint8), ("floats", C.c_float * 10)]


class TelemetryPacket(C.Structure):
    _fields_ = [("type", C.c_uint8), ("packet_number", C.c_uint8), ("accel", C.c_float * 3),
                ("gyro", C.c_float * 3), ("motor_forces", C.c_float * 4), ("position", C.c_float * 3),
                ("batt_voltage", C.c_float), ("velocity", C.c_float * 3), ("attitude", C.c_float * 3),
                ("debug_vals", C.c_float * 6), ("panic_reason", C.c_uint8), ("warnings", C.c_uint8)]


RADIO_PACKET_SIZE, TELEMETRY_PACKET_SIZE = 23, 30


class PlannerConfig(C.Structure):
    """afe_planner_config (RAPPIDS depth-image planner)"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("depth_scale", C.c_double), ("focal_length", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("true_vehicle_radius", C.c_double),
                ("planning_vehicle_radius", C.c_double), ("min_checking_dist", C.c_double),
                ("min_thrust", C.c_double), ("max_thrust", C.c_double), ("max_ang_vel", C.c_double),
                ("max_velocity", C.c_double), ("min_section_time", C.c_double), ("max_pyramids", C.c_int),
                ("pixel_buffer", C.c_int), ("cost_type", C.c_int), ("cost_vec", C.c_double * 3)]


class PlanOutput(C.Structure):
    _fields_ = [("found", C.c_int), ("best_index", C.c_int), ("best_cost", C.c_double),
                ("coeffs", (C.c_double * 3) * 6), ("tf", C.c_double), ("n_generated", C.c_int),
                ("n_cost_checks", C.c_int), ("n_collision_checks", C.c_int), ("n_velocity_checks", C.c_int),
                ("n_collision_free", C.c_int), ("n_pyramids", C.c_int)]


# PlanOutput as a numpy record (same layout), for hosts that consume 10^4..10^5 plans per frame
PLAN_DTYPE = np.dtype([("found", np.int32), ("best_index", np.int32), ("best_cost", np.float64),
                       ("coeffs", np.float64, (6, 3)), ("tf", np.float64), ("n_generated", np.int32),
                       ("n_cost_checks", np.int32), ("n_collision_checks", np.int32), ("n_velocity_checks", np.int32),
                       ("n_collision_free", np.int32), ("n_pyramids", np.int32)])


def plans_as_array(out):
    """zero-copy numpy view of the PlanOutput array rappids_plan returns"""
    assert PLAN_DTYPE.itemsize == C.sizeof(PlanOutput)
    return np.frombuffer(out, dtype=PLAN_DTYPE)


class DeviceView(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_size_t), ("n_vehicles", C.c_int64), ("stride", C.c_int64), ("state_elem_size", C.c_int),
        ("pos", C.c_void_p), ("vel", C.c_void_p), ("att", C.c_void_p),
        ("ang_vel", C.c_void_p), ("motor_speed", C.c_void_p),
        ("ext_force", C.c_void_p), ("ext_torque", C.c_void_p),
        ("motor_cmd", C.c_void_p), ("gyro", C.c_void_p), ("acc", C.c_void_p),
        ("rng", C.c_void_p), ("type_index", C.c_void_p), ("pos_anchor_xy", C.c_void_p),
    ]


def stream_probe(n, n_read=24, n_write=17, launches=100, device=-1):
    """afe_stream_probe: microseconds per launch of a pure streaming kernel in the step kernel's launch shape"""
    us = C.c_float(0)
    rc = library().afe_stream_probe(int(device), int(n), int(n_read), int(n_write), int(launches), C.byref(us))
    if rc:
        raise AfeError(rc, "afe_stream_probe")
    return us.value


def library_path():
    return _LIB


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ into lib/ (in-tree)."""
    src = os.path.join(_HERE, "csrc")
    if force and os.path.exists(_LIB):
        os.remove(_LIB)
    subprocess.check_call(["make", "-s", "-C", src])
    if not os.path.exists(_LIB):
        raise RuntimeError("engine build produced no " + _LIB)
    return _LIB


_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
_BC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
_GR = C.CFUNCTYPE(C.c_int, C.c_void_p)


class GatherTransport(C.Structure):
    """afe_gather_transport: the collectives afe_gather_exchange runs on"""
    _fields_ = [("ctx", C.c_void_p), ("all_gather", _AG), ("broadcast", _BC), ("group_start", _GR), ("group_end", _GR)]


def gather_exchange(all_gather, broadcast, rank, n_ranks, counts, packed_xyz, xyz_all):
    """afe_gather_exchange over host arrays with Python collectives:
    all_gather(send: float32[count], recv: float32[n_ranks * count]); broadcast(send, recv: float32[count], root).
    packed_xyz float32 [3, n_local], xyz_all float32 [3, n_all] (both C-contiguous, written in place)."""
    def view(ptr, count):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(int(count),))

    def ag(_, send, recv, count):
        try:
            all_gather(view(send, count), view(recv, count * n_ranks))
            return 0
        except Exception:      # noqa: BLE001  (an exception must not cross the C frame)
            return 1

    def bc(_, send, recv, count, root):
        try:
            broadcast(view(send, count), view(recv, count), int(root))
            return 0
        except Exception:      # noqa: BLE001
            return 1

    t = GatherTransport(None, _AG(ag), _BC(bc), _GR(), _GR())
    cnt = None if counts is None else np.ascontiguousarray(counts, dtype=np.int64)
    assert packed_xyz.dtype == np.float32 and xyz_all.dtype == np.float32 and packed_xyz.flags.c_contiguous and xyz_all.flags.c_contiguous
    rc = library().afe_gather_exchange(C.byref(t), int(rank), int(n_ranks), None if cnt is None else cnt.ctypes.data,
                                       int(packed_xyz.shape[1]), packed_xyz.ctypes.data, xyz_all.ctypes.data)
    if rc:
        raise AfeError(rc, "afe_gather_exchange")


class Camera(C.Structure):
    """afe_camera: the pinhole depth camera of Rappids_Simulator/main.cpp:120-122,360."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("focal_length", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("depth_scale", C.c_double),
        ("max_count", C.c_int32), ("reserved", C.c_int32),
    ]


_lib = None


def library():
    """Load the engine library; fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise ImportError(
            "HIP engine library %s is missing: run `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (hipcc, gfx950). There is no CPU fallback." % _LIB)
    L = C.CDLL(_LIB)
    i64, u64, vp, ci = C.c_int64, C.c_uint64, C.c_void_p, C.c_int
    eng = vp
    sig = {
        "afe_params_from_type": [ci, C.POINTER(VehicleParams)],
        "afe_type_from_id": [C.c_uint],
        "afe_create": [C.POINTER(vp), i64, ci, ci, i64],
        "afe_create_host_visible": [C.POINTER(vp), i64, ci, ci, i64],
        "afe_destroy": [eng],
        "afe_abi_version": [],
        "afe_set_stream": [eng, vp],
        "afe_set_type_table": [eng, C.POINTER(VehicleParams), ci],
        "afe_set_vehicle_types": [eng, i64, i64, vp],
        "afe_set_logic_period": [eng, C.c_double],
        "afe_set_imu_noise": [eng, ci, C.c_double, C.c_double, ci],
        "afe_set_state": [eng, i64, i64, vp, vp, vp, vp, vp],
        "afe_get_state": [eng, i64, i64, vp, vp, vp, vp, vp],
        "afe_set_state_f32": [eng, i64, i64, vp, vp, vp, vp, vp],
        "afe_get_state_f32": [eng, i64, i64, vp, vp, vp, vp, vp],
        "afe_set_rng_state": [eng, i64, i64, vp],
        "afe_get_rng_state": [eng, i64, i64, vp],
        "afe_set_motor_cmds": [eng, i64, i64, vp],
        "afe_set_external_force": [eng, i64, i64, vp],
        "afe_set_external_torque": [eng, i64, i64, vp],
        "afe_step": [eng, u64, ci],
        "afe_steps_until_tick": [eng, u64, C.POINTER(ci)],
        "afe_sync": [eng],
        "afe_time_us": [eng, C.POINTER(u64)],
        "afe_logic_ticks": [eng, C.POINTER(u64)],
        "afe_get_imu": [eng, i64, i64, vp, vp],
        "afe_plan_ticks": [C.c_double, C.POINTER(u64), u64, ci, vp],
        "afe_get_device_view": [eng, C.POINTER(DeviceView)],
        "afe_algorithmic_bytes_per_step": [eng, ci, C.POINTER(C.c_double)],
        "afe_event_create": [C.POINTER(vp)],
        "afe_event_destroy": [vp],
        "afe_event_record": [eng, vp],
        "afe_event_elapsed_ms": [vp, vp, C.POINTER(C.c_float)],
        "afe_pack_positions": [eng, vp],
        "afe_nearest_neighbour": [eng, vp, i64, vp, vp],
        "afe_selftest_normals": [eng, vp, i64, vp, vp],
        "afe_selftest_normals_f32": [eng, vp, i64, vp, vp],
        "afe_rates_logic_params_from_type": [ci, C.POINTER(RatesLogicParams)],
        "afe_set_rates_logic": [eng, C.POINTER(RatesLogicParams), ci],
        "afe_set_rates_commands": [eng, i64, i64, vp, vp],
        "afe_get_motor_cmds": [eng, i64, i64, vp],
        "afe_radio_create_rates_command": [C.c_uint8, C.c_float, vp, vp],
        "afe_radio_create_position_command": [C.c_uint8, vp, vp, vp, vp],
        "afe_radio_create_acceleration_command": [C.c_uint8, vp, C.c_float, vp],
        "afe_radio_create_simple_command": [ci, C.c_uint8, vp],
        "afe_radio_decode": [vp, C.POINTER(RadioMessage)],
        "afe_telemetry_encode": [C.POINTER(TelemetryPacket), vp],
        "afe_telemetry_decode": [vp, C.POINTER(TelemetryPacket)],
        "afe_set_commands_from_radio": [eng, i64, i64, vp],
        "afe_set_max_fused_steps": [eng, ci],
        "afe_set_split_stepping": [eng, ci],
        "afe_set_step_mode": [eng, ci],
        "afe_nearest_neighbour_async": [eng, vp, i64, vp, vp],
        "afe_query_sync": [eng],
        "afe_gather_exchange": [C.POINTER(GatherTransport), ci, ci, vp, i64, vp, vp],
        "afe_set_noise_seed": [eng, u64],
        "afe_set_gust_process": [eng, ci, u64, C.c_double, u64, i64],
        "afe_get_external_force": [eng, i64, i64, vp],
        "afe_stream_probe": [ci, i64, ci, ci, ci, C.POINTER(C.c_float)],
        "afe_steps_completed": [eng, C.POINTER(u64)],
        "afe_persistent_running": [eng, C.POINTER(ci)],
        "afe_set_addressing": [eng, ci],
        "afe_step_kernel_info": [eng, C.POINTER(ci), C.POINTER(ci)],
        "afe_planner_default_config": [C.POINTER(PlannerConfig), ci, ci] + [C.c_double] * 5,
        "afe_planner_samples": [C.c_uint32, ci, ci, ci, vp],
        "afe_rappids_plan": [ci, C.POINTER(PlannerConfig), i64, vp, i64, vp, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp,
                             C.POINTER(C.c_float)],
        "afe_planner_release_scratch": [],
        "afe_rappids_plan_device": [ci, C.POINTER(PlannerConfig), i64, vp, i64, vp, vp, vp, vp, vp, vp, ci, vp, ci, vp,
                                    vp, C.POINTER(C.c_float)],
        "afe_camera_default": [C.POINTER(Camera), ci, ci],
        "afe_camera_default_mount": [vp],
        "afe_scene_create": [ci, vp, i64, C.POINTER(vp)],
        "afe_scene_info": [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(ci), vp],
        "afe_scene_check_hierarchy": [vp, i64, C.POINTER(i64), C.POINTER(ci), C.POINTER(ci)],
        "afe_render_depth": [vp, C.POINTER(Camera), i64, vp, vp, vp, vp, C.POINTER(C.c_float)],
        "afe_render_depth_engine": [eng, vp, C.POINTER(Camera), i64, i64, vp, vp, ci, C.POINTER(C.c_float)],
        "afe_render_depth_stats": [vp, C.POINTER(Camera), i64, vp, vp, vp, vp, C.POINTER(C.c_float)],
        "afe_scene_set_walk": [vp, C.c_int],
        "afe_device_alloc": [ci, u64, C.POINTER(vp)],
        "afe_device_free": [vp],
        "afe_device_download": [vp, vp, u64],
        "afe_checkpoint_size": [eng, C.POINTER(u64)],
        "afe_save_checkpoint": [eng, vp, u64],
        "afe_load_checkpoint": [eng, vp, u64],
        "afe_comm_unique_id": [vp],
        "afe_comm_create": [C.POINTER(vp), vp, ci, ci, ci],
        "afe_comm_info": [vp, C.POINTER(ci), C.POINTER(ci)],
        "afe_comm_destroy": [vp],
        "afe_gather_positions": [eng, vp, vp, vp],
        "afe_group_create": [C.POINTER(vp), i64, ci, vp, ci],
        "afe_group_peer_access": [vp, C.POINTER(ci)],
        "afe_set_cache_policy": [eng, ci],
        "afe_cache_policy_in_use": [eng, C.POINTER(ci)],
        "afe_has_dev_hooks": [],
        "afe_persistent_kernarg_layout": [ci, vp, vp, vp],
        "afe_group_set_staged_copies": [vp, ci],
        "afe_set_resident_queue": [eng, ci],
        "afe_grid_time": [eng, C.POINTER(u64), C.POINTER(u64)],
        "afe_group_destroy": [vp],
        "afe_group_size": [vp, C.POINTER(ci), C.POINTER(i64)],
        "afe_group_shard": [vp, ci, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)],
        "afe_group_step": [vp, u64, ci],
        "afe_group_sync": [vp],
        "afe_group_gather_positions": [vp, vp],
        "afe_nearest_neighbour_grid": [eng, vp, i64, C.c_float, vp, vp],
        "afe_neighbour_grid_info": [eng, vp, C.POINTER(C.c_float), C.POINTER(i64), C.POINTER(i64)],
        "afe_nearest_neighbour_bruteforce": [eng, vp, i64, vp, i64, vp, vp],
        "afe_set_neighbour_grid_refresh": [eng, ci],
        "afe_set_neighbour_sort_reuse": [eng, ci],
        "afe_uwb_create": [C.POINTER(vp)],
        "afe_uwb_set_noise": [vp, C.c_double, C.c_double, C.c_double],
        "afe_uwb_draw": [vp, i64, vp, vp],
        "afe_uwb_range": [vp, eng, vp, i64, vp, vp, i64, vp, vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = ci
    L.afe_scene_destroy.argtypes = [vp]
    L.afe_scene_destroy.restype = None
    L.afe_uwb_destroy.argtypes = [vp]
    L.afe_uwb_destroy.restype = None
    for name in ("afe_comm_last_error", "afe_group_last_error"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.c_char_p
    L.afe_last_error.argtypes = [eng]
    L.afe_last_error.restype = C.c_char_p
    L.afe_status_string.argtypes = [ci]
    L.afe_status_string.restype = C.c_char_p
    _lib = L
    return L


def params_from_type(quadcopter_type):
    p = VehicleParams()
    rc = library().afe_params_from_type(int(quadcopter_type), C.byref(p))
    if rc:
        raise AfeError(rc, "invalid quadcopter type %r" % (quadcopter_type,))
    return p


def rates_logic_params_from_type(quadcopter_type):
    p = RatesLogicParams()
    rc = library().afe_rates_logic_params_from_type(int(quadcopter_type), C.byref(p))
    if rc:
        raise AfeError(rc, "invalid quadcopter type %r" % (quadcopter_type,))
    return p


def radio_create_rates_command(flags, thrust, ang_vel):
    raw = np.zeros(RADIO_PACKET_SIZE, np.uint8)
    w = np.ascontiguousarray(ang_vel, dtype=np.float32)
    rc = library().afe_radio_create_rates_command(int(flags), float(thrust), w.ctypes.data, raw.ctypes.data)
    if rc:
        raise AfeError(rc, "afe_radio_create_rates_command")
    return raw


def radio_decode(raw):
    r = np.ascontiguousarray(raw, dtype=np.uint8)
    m = RadioMessage()
    rc = library().afe_radio_decode(r.ctypes.data, C.byref(m))
    if rc:
        raise AfeError(rc, "afe_radio_decode")
    return m


def planner_default_config(width, height, depth_scale, focal_length, true_radius, planning_radius,
                           min_checking_dist):
    c = PlannerConfig()
    rc = library().afe_planner_default_config(C.byref(c), width, height, depth_scale, focal_length, true_radius,
                                              planning_radius, min_checking_dist)
    if rc:
        raise AfeError(rc, "afe_planner_default_config")
    return c


def planner_samples(seed, width, height, n_candidates):
    s = np.empty((n_candidates, 4))
    rc = library().afe_planner_samples(int(seed), width, height, n_candidates, s.ctypes.data)
    if rc:
        raise AfeError(rc, "afe_planner_samples")
    return s


def planner_release_scratch():
    """afe_planner_release_scratch: give back the device scratch the planner keeps between calls"""
    library().afe_planner_release_scratch()


def rappids_plan(cfg, depth_images, vel0, acc0, grav, samples, image_index=None, cost_vec=None,
                 sample_table=None, want_flags=False, device=-1):
    """Batched RAPPIDS plan.  depth_images uint16 [n_images, H, W] (or a DeviceBuffer holding them,
    e.g. filled by Scene.render_engine); vel0/acc0/grav [3, n];
    samples [n_tables, M, 4] (or [M, 4]).  Returns (PlanOutput array, flags or None, kernel_ms)."""
    on_device = isinstance(depth_images, DeviceBuffer)
    if not on_device:
        img = np.ascontiguousarray(depth_images, dtype=np.uint16)
        if img.ndim == 2:
            img = img[None]
    v = np.ascontiguousarray(vel0, dtype=np.float64)
    n = v.shape[1]
    a = np.ascontiguousarray(acc0, dtype=np.float64)
    g = np.ascontiguousarray(grav, dtype=np.float64)
    s = np.ascontiguousarray(samples, dtype=np.float64)
    if s.ndim == 2:
        s = s[None]
    assert v.shape == a.shape == g.shape == (3, n) and s.shape[2] == 4
    if on_device:
        n_images = depth_images.nbytes // (2 * cfg.height * cfg.width)
        img_ptr, fn = depth_images.ptr, library().afe_rappids_plan_device
    else:
        assert img.shape[1:] == (cfg.height, cfg.width)
        n_images, img_ptr, fn = img.shape[0], img.ctypes.data, library().afe_rappids_plan
    idx = None if image_index is None else np.ascontiguousarray(image_index, dtype=np.int32)
    cv = None if cost_vec is None else np.ascontiguousarray(cost_vec, dtype=np.float64)
    st = None if sample_table is None else np.ascontiguousarray(sample_table, dtype=np.int32)
    out = (PlanOutput * n)()
    flags = np.zeros((n, s.shape[1]), np.uint8) if want_flags else None
    ms = C.c_float(0)
    rc = fn(int(device), C.byref(cfg), n, img_ptr, n_images,
            None if idx is None else idx.ctypes.data, v.ctypes.data, a.ctypes.data,
            g.ctypes.data, None if cv is None else cv.ctypes.data, s.ctypes.data, s.shape[0],
            None if st is None else st.ctypes.data, s.shape[1], out,
            None if flags is None else flags.ctypes.data, C.byref(ms))
    if rc:
        raise AfeError(rc, library().afe_status_string(rc).decode())
    return out, flags, ms.value


def camera_default(width, height):
    cam = Camera()
    rc = library().afe_camera_default(C.byref(cam), int(width), int(height))
    if rc:
        raise AfeError(rc, "afe_camera_default")
    return cam


def camera_default_mount():
    """depthCamAtt (main.cpp:123-125) as (w, x, y, z)."""
    q = np.empty(4)
    rc = library().afe_camera_default_mount(q.ctypes.data)
    if rc:
        raise AfeError(rc, "afe_camera_default_mount")
    return q


class DeviceBuffer:
    """A raw HBM allocation (afe_device_alloc) for chaining render -> plan on the device."""

    def __init__(self, nbytes, device=-1):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        rc = library().afe_device_alloc(int(device), self.nbytes, C.byref(self.ptr))
        if rc:
            raise AfeError(rc, library().afe_status_string(rc).decode())

    def download(self, dtype, shape, offset_bytes=0):
        out = np.empty(shape, dtype)
        assert offset_bytes >= 0 and offset_bytes + out.nbytes <= self.nbytes
        rc = library().afe_device_download(out.ctypes.data, C.c_void_p(self.ptr.value + int(offset_bytes)), out.nbytes)
        if rc:
            raise AfeError(rc, library().afe_status_string(rc).decode())
        return out

    def close(self):
        if self.ptr:
            library().afe_device_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """afe_scene: a static triangle mesh (world frame, metres) + its BVH in HBM."""

    def __init__(self, triangles, device=-1):
        t = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
        self._h = C.c_void_p()
        rc = library().afe_scene_create(int(device), t.ctypes.data, t.shape[0], C.byref(self._h))
        if rc:
            raise AfeError(rc, library().afe_status_string(rc).decode())

    def close(self):
        if self._h:
            library().afe_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        nt, nn, d = C.c_int64(), C.c_int64(), C.c_int()
        b = np.empty(6)
        rc = library().afe_scene_info(self._h, C.byref(nt), C.byref(nn), C.byref(d), b.ctypes.data)
        if rc:
            raise AfeError(rc, "afe_scene_info")
        return {"n_tri": nt.value, "n_nodes": nn.value, "depth": d.value, "bounds": b}

    def render(self, cam, pos, att, mount=None):
        """pos [3, n], att [4, n] -> (uint16 [n, H, W], kernel_ms)."""
        p = np.ascontiguousarray(pos, dtype=np.float64)
        q = np.ascontiguousarray(att, dtype=np.float64)
        n = p.shape[1]
        assert p.shape == (3, n) and q.shape == (4, n)
        m = None if mount is None else np.ascontiguousarray(mount, dtype=np.float64)
        out = np.empty((n, cam.height, cam.width), np.uint16)
        ms = C.c_float(0)
        rc = library().afe_render_depth(self._h, C.byref(cam), n, p.ctypes.data, q.ctypes.data,
                                        None if m is None else m.ctypes.data, out.ctypes.data, C.byref(ms))
        if rc:
            raise AfeError(rc, library().afe_status_string(rc).decode())
        return out, ms.value

    def set_walk(self, plain_only):
        """False (default): ordered walk of the octant-mirrored trees where a tile allows it; True: the plain
        walk everywhere.  Same images -- a cross-check."""
        rc = library().afe_scene_set_walk(self._h, 1 if plain_only else 0)
        if rc:
            raise AfeError(rc, library().afe_status_string(rc).decode())

    def render_stats(self, cam, pos, att, mount=None):
        """traversal counters of one batch (counting build): dict + kernel_ms"""
        p = np.ascontiguousarray(pos, dtype=np.float64)
        q = np.ascontiguousarray(att, dtype=np.float64)
        n = p.shape[1]
        m = None if mount is None else np.ascontiguousarray(mount, dtype=np.float64)
        st = np.zeros(8, np.uint64)
        ms = C.c_float(0)
        rc = library().afe_render_depth_stats(self._h, C.byref(cam), n, p.ctypes.data, q.ctypes.data,
                                              None if m is None else m.ctypes.data, st.ctypes.data, C.byref(ms))
        if rc:
            raise AfeError(rc, library().afe_status_string(rc).decode())
        keys = ("nodes_per_wave", "tri_box_tests_per_wave", "tri_fp64_tests_per_wave", "tri_box_tests_per_ray",
                "tri_fp64_tests_per_ray", "rays", "waves", "visible_triangles_per_wave")
        return dict(zip(keys, (int(x) for x in st[:8]))), ms.value

    def render_engine(self, ensemble, cam, mount=None, first=0, count=None, out=None):
        """Depth images of vehicles [first, first+count) from the engine's device state.
        out: a DeviceBuffer to keep the images in HBM (returns kernel_ms only), or None to
        get (uint16 [count, H, W], kernel_ms) on the host."""
        count = ensemble.n - first if count is None else count
        m = None if mount is None else np.ascontiguousarray(mount, dtype=np.float64)
        ms = C.c_float(0)
        if out is None:
            img = np.empty((count, cam.height, cam.width), np.uint16)
            dst, is_dev = img.ctypes.data, 0
        else:
            assert out.nbytes >= count * cam.height * cam.width * 2
            img, dst, is_dev = None, out.ptr, 1
        rc = library().afe_render_depth_engine(ensemble.handle, self._h, C.byref(cam), first, count,
                                               None if m is None else m.ctypes.data, dst, is_dev, C.byref(ms))
        if rc:
            raise AfeError(rc, library().afe_status_string(rc).decode())
        return (img, ms.value) if out is None else ms.value


def scene_check_hierarchy(triangles):
    """Host-only: (n_nodes, depth, max_leaf) of the BVH for a mesh, after verifying its invariants."""
    t = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
    nn, d, ml = C.c_int64(), C.c_int(), C.c_int()
    rc = library().afe_scene_check_hierarchy(t.ctypes.data, t.shape[0], C.byref(nn), C.byref(d), C.byref(ml))
    if rc:
        raise AfeError(rc, library().afe_status_string(rc).decode())
    return nn.value, d.value, ml.value


def type_from_id(vehicle_id):
    return library().afe_type_from_id(int(vehicle_id))


def plan_ticks(logic_period, elapsed_us, dt_us, n_steps):
    """(ticks[n_steps], new elapsed_us): pure host, no GPU needed."""
    el = C.c_uint64(int(elapsed_us))
    out = np.zeros(n_steps, np.uint8)
    rc = library().afe_plan_ticks(float(logic_period), C.byref(el), int(dt_us), int(n_steps),
                                  out.ctypes.data)
    if rc:
        raise AfeError(rc, "afe_plan_ticks")
    return out, el.value


def _planar(a, comps, count, dtype):
    if a is None:
        return None, None
    arr = np.ascontiguousarray(a, dtype=dtype)
    if arr.shape != (comps, count):
        raise ValueError("expected planar array of shape (%d, %d), got %r" % (comps, count, arr.shape))
    return arr, arr.ctypes.data


class Ensemble:
    """One engine = one vehicle ensemble (or one rank's shard of it) on one GPU."""

    def __init__(self, n_vehicles, precision=AFE_F32, device=-1, first_global_index=0, _borrowed=None, host_visible=False):
        """host_visible: afe_create_host_visible -- the state arena in pinned host memory (small ensembles with the host in
        the loop of every step: getters and setters become host copies and leave a resident grid where it is)."""
        self._L = library()
        self._owned = _borrowed is None
        if _borrowed is not None:   # a shard of a Group: the group owns the engine
            self._h = _borrowed
            self.n = int(n_vehicles)
            self.precision = precision
            self.first_global_index = int(first_global_index)
            return
        self._h = C.c_void_p()
        create = self._L.afe_create_host_visible if host_visible else self._L.afe_create
        rc = create(C.byref(self._h), int(n_vehicles), int(precision), int(device), int(first_global_index))
        if rc:
            self._h = None
            raise AfeError(rc, self._L.afe_status_string(rc).decode() +
                           " (the HIP engine needs an MI355X / gfx950 device; there is no CPU fallback)")
        self.n = int(n_vehicles)
        self.precision = precision
        self.first_global_index = int(first_global_index)

    # -- plumbing ---------------------------------------------------------
    def _ck(self, rc):
        if rc:
            raise AfeError(rc, self._L.afe_last_error(self._h).decode() or
                           self._L.afe_status_string(rc).decode())

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._L.afe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    def _range(self, first, count):
        first = int(first)
        count = self.n - first if count is None else int(count)
        return first, count

    # -- configuration ----------------------------------------------------
    def set_type_table(self, params_list):
        arr = (VehicleParams * len(params_list))()
        for i, p in enumerate(params_list):
            C.memmove(C.byref(arr[i]), C.byref(p), C.sizeof(VehicleParams))
        self._ck(self._L.afe_set_type_table(self._h, arr, len(params_list)))

    def set_vehicle_types(self, type_index, first=0):
        t = np.ascontiguousarray(type_index, dtype=np.uint8)
        self._ck(self._L.afe_set_vehicle_types(self._h, int(first), t.size, t.ctypes.data))

    def set_logic_period(self, seconds):
        self._ck(self._L.afe_set_logic_period(self._h, float(seconds)))

    def set_imu_noise(self, enabled=True, sigma_gyro=0.1, sigma_acc=0.2,
                      seed_policy=AFE_SEED_REFERENCE):
        self._ck(self._L.afe_set_imu_noise(self._h, int(bool(enabled)), float(sigma_gyro),
                                           float(sigma_acc), int(seed_policy)))

    def set_stream(self, hip_stream):
        self._ck(self._L.afe_set_stream(self._h, C.c_void_p(hip_stream or None)))

    # -- state ------------------------------------------------------------
    def set_state(self, pos=None, vel=None, att=None, ang_vel=None, motor_speed=None,
                  first=0, count=None, dtype=np.float64):
        first, count = self._range(first, count)
        fn = self._L.afe_set_state if dtype == np.float64 else self._L.afe_set_state_f32
        keep = [_planar(a, c, count, dtype) for a, c in
                ((pos, 3), (vel, 3), (att, 4), (ang_vel, 3), (motor_speed, 4))]
        self._ck(fn(self._h, first, count, *[k[1] for k in keep]))

    def get_state(self, first=0, count=None, dtype=np.float64):
        first, count = self._range(first, count)
        fn = self._L.afe_get_state if dtype == np.float64 else self._L.afe_get_state_f32
        out = dict(pos=np.empty((3, count), dtype), vel=np.empty((3, count), dtype),
                   att=np.empty((4, count), dtype), ang_vel=np.empty((3, count), dtype),
                   motor_speed=np.empty((4, count), dtype))
        self._ck(fn(self._h, first, count, out["pos"].ctypes.data, out["vel"].ctypes.data,
                    out["att"].ctypes.data, out["ang_vel"].ctypes.data,
                    out["motor_speed"].ctypes.data))
        return out

    def set_rng_state(self, state, first=0):
        s = np.ascontiguousarray(state, dtype=np.uint32)
        self._ck(self._L.afe_set_rng_state(self._h, int(first), s.size, s.ctypes.data))

    def get_rng_state(self, first=0, count=None):
        first, count = self._range(first, count)
        s = np.empty(count, np.uint32)
        self._ck(self._L.afe_get_rng_state(self._h, first, count, s.ctypes.data))
        return s

    # -- inputs -----------------------------------------------------------
    def set_motor_cmds(self, cmd4, first=0, count=None):
        first, count = self._range(first, count)
        arr, ptr = _planar(cmd4, 4, count, np.float32)
        self._ck(self._L.afe_set_motor_cmds(self._h, first, count, ptr))

    def set_external_force(self, force3, first=0, count=None):
        first, count = self._range(first, count)
        arr, ptr = _planar(force3, 3, count, np.float64)
        self._ck(self._L.afe_set_external_force(self._h, first, count, ptr))

    def set_external_torque(self, torque3, first=0, count=None):
        first, count = self._range(first, count)
        arr, ptr = _planar(torque3, 3, count, np.float64)
        self._ck(self._L.afe_set_external_torque(self._h, first, count, ptr))

    # -- on-device onboard rates logic (f1) ---------------------------------
    def set_rates_logic(self, params_list):
        """enable (list of RatesLogicParams, one per vehicle type) or disable (None)"""
        if params_list is None:
            self._ck(self._L.afe_set_rates_logic(self._h, None, 0))
            return
        arr = (RatesLogicParams * len(params_list))()
        for i, p in enumerate(params_list):
            C.memmove(C.byref(arr[i]), C.byref(p), C.sizeof(RatesLogicParams))
        self._ck(self._L.afe_set_rates_logic(self._h, arr, len(params_list)))

    def set_rates_commands(self, thrust_norm, ang_vel3, first=0, count=None):
        first, count = self._range(first, count)
        t = np.ascontiguousarray(thrust_norm, dtype=np.float32)
        if t.shape != (count,):
            raise ValueError("thrust_norm must have shape (%d,)" % count)
        w, wp = _planar(ang_vel3, 3, count, np.float32)
        self._ck(self._L.afe_set_rates_commands(self._h, first, count, t.ctypes.data, wp))

    def set_commands_from_radio(self, raw_packets, first=0):
        """raw_packets: uint8 [count, 23]"""
        r = np.ascontiguousarray(raw_packets, dtype=np.uint8)
        if r.ndim != 2 or r.shape[1] != RADIO_PACKET_SIZE:
            raise ValueError("raw_packets must have shape (count, 23)")
        self._ck(self._L.afe_set_commands_from_radio(self._h, int(first), r.shape[0], r.ctypes.data))

    def get_motor_cmds(self, first=0, count=None):
        first, count = self._range(first, count)
        out = np.empty((4, count), np.float32)
        self._ck(self._L.afe_get_motor_cmds(self._h, first, count, out.ctypes.data))
        return out

    # -- stepping ---------------------------------------------------------
    def step(self, dt_us, n_steps=1):
        self._ck(self._L.afe_step(self._h, int(dt_us), int(n_steps)))

    def set_addressing(self, force_global):
        """False (default): buffer resources when the arenas fit 32-bit offsets; True: global addresses always"""
        self._ck(self._L.afe_set_addressing(self._h, 1 if force_global else 0))

    def step_kernel_info(self):
        """(record path, addressing): ("kernel arguments" | "per-wave scalar loads" | "LDS table", "buffer" | "global")"""
        a, b = C.c_int(0), C.c_int(0)
        self._ck(self._L.afe_step_kernel_info(self._h, C.byref(a), C.byref(b)))
        return ("kernel arguments", "per-wave scalar loads", "LDS table")[a.value], ("buffer", "global")[b.value]

    def set_max_fused_steps(self, k):
        self._ck(self._L.afe_set_max_fused_steps(self._h, int(k)))

    def set_split_stepping(self, parts):
        """afe_set_split_stepping: 0 automatic (default), 1 off, 2 = the two halves of the ensemble step on two streams (see the header)"""
        self._ck(self._L.afe_set_split_stepping(self._h, int(parts)))

    def set_noise_seed(self, seed):
        self._ck(self._L.afe_set_noise_seed(self._h, int(seed)))

    def set_gust_process(self, enabled, seed=0, sigma_max=0.5, period_us=100000, n_global=0):
        """afe_set_gust_process: per-vehicle piecewise-constant wind gusts resampled on the device (BASELINE config 4)"""
        self._ck(self._L.afe_set_gust_process(self._h, int(bool(enabled)), int(seed), float(sigma_max), int(period_us), int(n_global)))

    def get_external_force(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.empty((3, count), np.float64)
        self._ck(self._L.afe_get_external_force(self._h, int(first), int(count), out.ctypes.data))
        return out

    def set_step_mode(self, mode):
        """afe_set_step_mode: AFE_STEP_LAUNCH (0), AFE_STEP_PERSISTENT (1: one resident grid, afe_step only authorises steps), AFE_STEP_AUTO (2)"""
        self._ck(self._L.afe_set_step_mode(self._h, int(mode)))

    @property
    def steps_completed(self):
        """steps every vehicle has been advanced through (the resident grid's completion word; steps issued in launch mode)"""
        n = C.c_uint64(0)
        self._ck(self._L.afe_steps_completed(self._h, C.byref(n)))
        return n.value

    @property
    def persistent_running(self):
        r = C.c_int(0)
        self._ck(self._L.afe_persistent_running(self._h, C.byref(r)))
        return bool(r.value)

    def steps_until_tick(self, dt_us):
        n = C.c_int(0)
        self._ck(self._L.afe_steps_until_tick(self._h, int(dt_us), C.byref(n)))
        return n.value

    def sync(self):
        self._ck(self._L.afe_sync(self._h))

    @property
    def time_us(self):
        t = C.c_uint64(0)
        self._ck(self._L.afe_time_us(self._h, C.byref(t)))
        return t.value

    @property
    def logic_ticks(self):
        t = C.c_uint64(0)
        self._ck(self._L.afe_logic_ticks(self._h, C.byref(t)))
        return t.value

    def get_imu(self, first=0, count=None):
        first, count = self._range(first, count)
        gyro = np.empty((3, count), np.float32)
        acc = np.empty((3, count), np.float32)
        self._ck(self._L.afe_get_imu(self._h, first, count, gyro.ctypes.data, acc.ctypes.data))
        return gyro, acc

    def device_view(self):
        v = DeviceView()
        v.struct_bytes = C.sizeof(DeviceView)
        self._ck(self._L.afe_get_device_view(self._h, C.byref(v)))
        return v

    def grid_time(self):
        """(device seconds, steps) of the resident grids since the last call; ends the grid now resident"""
        ns, steps = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.afe_grid_time(self._h, C.byref(ns), C.byref(steps)))
        return ns.value * 1e-9, steps.value

    def set_cache_policy(self, policy):
        """-1 automatic, 0 default, 1 inputs / outputs nt, 2 everything nt, 3 everything nt + one range per XCD"""
        self._ck(self._L.afe_set_cache_policy(self._h, int(policy)))

    def set_resident_queue(self, mode):
        """-1 automatic (own queue up to 262 144 vehicles), 0 the HIP stream, 1 the engine's own queue"""
        self._ck(self._L.afe_set_resident_queue(self._h, int(mode)))

    @property
    def cache_policy_in_use(self):
        p = C.c_int(0)
        self._ck(self._L.afe_cache_policy_in_use(self._h, C.byref(p)))
        return p.value

    def algorithmic_bytes_per_step(self, imu_tick):
        b = C.c_double(0)
        self._ck(self._L.afe_algorithmic_bytes_per_step(self._h, int(bool(imu_tick)), C.byref(b)))
        return b.value

    # -- checkpoint / resume --------------------------------------------------
    CHECKPOINT_HEADER_BYTES = 27 * 8    # CheckpointHeader (afe_engine.cpp): 17 x uint64, 3 x double, 6 x uint64, 1 x double; the arena follows

    def save_checkpoint(self):
        n = C.c_uint64(0)
        self._ck(self._L.afe_checkpoint_size(self._h, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        self._ck(self._L.afe_save_checkpoint(self._h, buf.ctypes.data, n.value))
        return buf

    def load_checkpoint(self, buf):
        b = np.ascontiguousarray(buf, dtype=np.uint8)
        self._ck(self._L.afe_load_checkpoint(self._h, b.ctypes.data, b.size))

    # -- HIP events on the engine stream ------------------------------------
    def event(self):
        ev = C.c_void_p()
        rc = self._L.afe_event_create(C.byref(ev))
        if rc:
            raise AfeError(rc, "afe_event_create")
        return ev

    def record(self, ev):
        self._ck(self._L.afe_event_record(self._h, ev))

    def elapsed_ms(self, start, stop):
        ms = C.c_float(0)
        rc = self._L.afe_event_elapsed_ms(start, stop, C.byref(ms))
        if rc:
            raise AfeError(rc, "afe_event_elapsed_ms")
        return ms.value

    def destroy_event(self, ev):
        self._L.afe_event_destroy(ev)

    def selftest_normals(self, seeds, dtype=np.float64):
        """(normals[n, 6], state_after[n] uint32) from the device generator: float64 = the AFE_F64
        engine's (libstdc++'s doubles), float32 = the AFE_F32 engine's (float multiplier)"""
        s = np.ascontiguousarray(seeds, dtype=np.uint32)
        out = np.empty((s.size, 6), dtype)
        st = np.empty(s.size, np.uint32)
        fn = self._L.afe_selftest_normals if dtype == np.float64 else self._L.afe_selftest_normals_f32
        self._ck(fn(self._h, s.ctypes.data, s.size, out.ctypes.data, st.ctypes.data))
        return out, st

    # -- shared-world exchange and queries ------------------------------------
    def pack_positions(self, device_ptr):
        self._ck(self._L.afe_pack_positions(self._h, C.c_void_p(int(device_ptr))))

    def gather_positions(self, comm, out_ptr, counts=None):
        """pack + RCCL all-gather on the engine's stream into a device buffer of 3*n_all floats"""
        cnt = None if counts is None else np.ascontiguousarray(counts, dtype=np.int64)
        rc = self._L.afe_gather_positions(self._h, comm.handle, None if cnt is None else cnt.ctypes.data,
                                          C.c_void_p(int(out_ptr)))
        if rc:
            raise AfeError(rc, (self._L.afe_comm_last_error(comm.handle) or b"").decode() or
                           self._L.afe_status_string(rc).decode())

    def nearest_neighbour(self, all_xyz_ptr, n_all, dist2_ptr, index_ptr, cell_size=0.0):
        self._ck(self._L.afe_nearest_neighbour_grid(self._h, C.c_void_p(int(all_xyz_ptr)), int(n_all), float(cell_size),
                                                    C.c_void_p(int(dist2_ptr)), C.c_void_p(int(index_ptr))))

    def nearest_neighbour_async(self, all_xyz_ptr, n_all, dist2_ptr, index_ptr):
        """afe_nearest_neighbour_async: the query on its own stream behind the gather; later steps are not ordered behind it"""
        self._ck(self._L.afe_nearest_neighbour_async(self._h, C.c_void_p(int(all_xyz_ptr)), int(n_all),
                                                     C.c_void_p(int(dist2_ptr)), C.c_void_p(int(index_ptr))))

    def query_sync(self):
        self._ck(self._L.afe_query_sync(self._h))

    def nearest_neighbour_bruteforce(self, all_xyz_ptr, n_all, queries_ptr, n_queries, dist2_ptr, index_ptr):
        self._ck(self._L.afe_nearest_neighbour_bruteforce(self._h, C.c_void_p(int(all_xyz_ptr)), int(n_all),
                                                          C.c_void_p(int(queries_ptr)), int(n_queries),
                                                          C.c_void_p(int(dist2_ptr)), C.c_void_p(int(index_ptr))))

    def set_neighbour_grid_refresh(self, every_n_queries):
        self._ck(self._L.afe_set_neighbour_grid_refresh(self._h, int(every_n_queries)))

    def set_neighbour_sort_reuse(self, every_n_queries):
        """sort the vehicles into the grid's cells every n-th query only; in between the order is kept and the positions
        refreshed (exact all the same: the device bounds the movement since the sort)"""
        self._ck(self._L.afe_set_neighbour_sort_reuse(self._h, int(every_n_queries)))

    def neighbour_grid_info(self):
        dims = (C.c_int * 3)()
        h, nc, nb = C.c_float(0), C.c_int64(0), C.c_int64(0)
        self._ck(self._L.afe_neighbour_grid_info(self._h, dims, C.byref(h), C.byref(nc), C.byref(nb)))
        return {"dims": tuple(dims), "cell_size": h.value, "n_cells": nc.value, "n_bruteforce": nb.value}


class Comm:
    """afe_comm: an RCCL communicator for the one-process-per-GPU layout."""

    @staticmethod
    def unique_id():
        uid = np.zeros(128, np.uint8)
        rc = library().afe_comm_unique_id(uid.ctypes.data)
        if rc:
            raise AfeError(rc, "afe_comm_unique_id (is RCCL loadable?)")
        return uid

    def __init__(self, unique_id, rank, n_ranks, device=-1):
        self._h = C.c_void_p()
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert uid.size == 128
        rc = library().afe_comm_create(C.byref(self._h), uid.ctypes.data, int(rank), int(n_ranks), int(device))
        if rc:
            self._h = None
            raise AfeError(rc, "afe_comm_create: " + library().afe_status_string(rc).decode())

    @property
    def handle(self):
        return self._h

    def info(self):
        r, n = C.c_int(0), C.c_int(0)
        rc = library().afe_comm_info(self._h, C.byref(r), C.byref(n))
        if rc:
            raise AfeError(rc, "afe_comm_info")
        return r.value, n.value

    def close(self):
        if getattr(self, "_h", None):
            library().afe_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    """afe_group: one process, several devices (or logical shards of one device)."""

    def __init__(self, n_vehicles, precision=AFE_F32, devices=(0,)):
        self._L = library()
        self._h = C.c_void_p()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        rc = self._L.afe_group_create(C.byref(self._h), int(n_vehicles), int(precision), dev.ctypes.data, dev.size)
        if rc:
            self._h = None
            raise AfeError(rc, self._L.afe_status_string(rc).decode())
        self.n = int(n_vehicles)
        self.shards = []
        for k in range(dev.size):
            e, first, count = C.c_void_p(), C.c_int64(0), C.c_int64(0)
            self._ck(self._L.afe_group_shard(self._h, k, C.byref(e), C.byref(first), C.byref(count)))
            self.shards.append(Ensemble(count.value, precision, first_global_index=first.value, _borrowed=e))

    def _ck(self, rc):
        if rc:
            raise AfeError(rc, (self._L.afe_group_last_error(self._h) or b"").decode() or
                           self._L.afe_status_string(rc).decode())

    def ranges(self):
        return [(s.first_global_index, s.n) for s in self.shards]

    def peer_access(self):
        """True if every pair of the group's devices reads the other's memory directly"""
        ok = C.c_int(0)
        self._ck(self._L.afe_group_peer_access(self._h, C.byref(ok)))
        return bool(ok.value)

    def set_staged_copies(self, staged):
        """the gather's staged path (hipMemcpyPeerAsync row by row) even between peers; False: direct copies again"""
        self._ck(self._L.afe_group_set_staged_copies(self._h, 1 if staged else 0))

    def step(self, dt_us, n_steps=1):
        self._ck(self._L.afe_group_step(self._h, int(dt_us), int(n_steps)))

    def sync(self):
        self._ck(self._L.afe_group_sync(self._h))

    def gather_positions(self):
        """device pointers (one per shard) of the planar fp32 [3][n] gathered positions"""
        ptrs = (C.c_void_p * len(self.shards))()
        self._ck(self._L.afe_group_gather_positions(self._h, ptrs))
        return [int(p) for p in ptrs]

    def close(self):
        if getattr(self, "_h", None):
            for s in self.shards:
                s.close()
            self._L.afe_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class UwbNetwork:
    """afe_uwb_network == Simulation::UWBNetwork (UWBNetwork.cpp:8-89), batched."""

    def __init__(self, noise_std=0.0, outlier_prob=0.0, outlier_std=0.0):
        self._h = C.c_void_p()
        rc = library().afe_uwb_create(C.byref(self._h))
        if rc:
            raise AfeError(rc, "afe_uwb_create")
        library().afe_uwb_set_noise(self._h, float(noise_std), float(outlier_prob), float(outlier_std))

    def set_noise(self, noise_std, outlier_prob, outlier_std):
        library().afe_uwb_set_noise(self._h, float(noise_std), float(outlier_prob), float(outlier_std))

    def draw(self, n_pairs):
        """the stream alone: (noise_term float64[n], is_outlier uint8[n]); host only"""
        noise = np.empty(n_pairs, np.float64)
        out = np.empty(n_pairs, np.uint8)
        rc = library().afe_uwb_draw(self._h, int(n_pairs), noise.ctypes.data, out.ctypes.data)
        if rc:
            raise AfeError(rc, "afe_uwb_draw")
        return noise, out

    def range(self, ensemble, all_xyz_ptr, n_all, requester, responder):
        req = np.ascontiguousarray(requester, dtype=np.int32)
        res = np.ascontiguousarray(responder, dtype=np.int32)
        assert req.shape == res.shape and req.ndim == 1
        rng = np.empty(req.size, np.float32)
        out = np.empty(req.size, np.uint8)
        rc = library().afe_uwb_range(self._h, ensemble.handle, C.c_void_p(int(all_xyz_ptr)), int(n_all), req.ctypes.data,
                                     res.ctypes.data, req.size, rng.ctypes.data, out.ctypes.data)
        if rc:
            raise AfeError(rc, library().afe_status_string(rc).decode())
        return rng, out

    def close(self):
        if getattr(self, "_h", None):
            library().afe_uwb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
