"""agri-fly_amd -- MI355X-native batched quadrotor dynamics engine.

This package is only the thin Python host over the C ABI of
``include/agrifly_engine.h`` (ctypes, numpy); the product is the HIP library
``lib/libagrifly_engine.so`` built from ``csrc/``.  There is no CPU fallback:
importing works anywhere (the library loads without a GPU so its host-only
entry points and symbol table can be checked), but creating an ``Ensemble``
without a gfx950 device raises.

The directory name carries a hyphen, so import it with
``importlib.import_module("agri-fly_amd")``.
"""
from .engine import (  # noqa: F401
    ABI_FUNCTIONS,
    AFE_F32,
    AFE_F64,
    AFE_SEED_COUNTER,
    AFE_SEED_DECORRELATED,
    AFE_SEED_REFERENCE,
    AFE_STEP_AUTO,
    AFE_STEP_LAUNCH,
    AFE_STEP_PERSISTENT,
    AFE_STEP_RESIDENT,
    AfeError,
    Camera,
    Comm,
    DeviceBuffer,
    DeviceView,
    Ensemble,
    Group,
    PlanOutput,
    PlannerConfig,
    RADIO_PACKET_SIZE,
    RadioMessage,
    RatesLogicParams,
    Scene,
    TELEMETRY_PACKET_SIZE,
    TelemetryPacket,
    UwbNetwork,
    VehicleParams,
    build_library,
    camera_default,
    gather_exchange,
    camera_default_mount,
    library,
    library_path,
    params_from_type,
    plan_ticks,
    planner_default_config,
    planner_release_scratch,
    planner_samples,
    plans_as_array,
    rappids_plan,
    radio_create_rates_command,
    radio_decode,
    rates_logic_params_from_type,
    scene_check_hierarchy,
    stream_probe,
    type_from_id,
)
from . import scenarios  # noqa: F401
from . import sharding  # noqa: F401
