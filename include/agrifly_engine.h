/*
 * agrifly_engine.h -- C ABI of the MI355X batched quadrotor dynamics engine.
 *
 * This is the drop-in boundary for agri-fly's vehicle-step hot path.  The
 * reference has no FFI: its seams are a C++ abstract class and a template
 * parameter compiled in-process.  The entry points below are
 * what a binding of that path has to reach; each one names the reference
 * interface it replaces (paths relative to the agri-fly tree).  The C++ facade
 * in include/agrifly/ re-creates the reference's Vehicle / logicType classes
 * on top of exactly these calls.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ or torch types.
 *   - every call returns an afe_status (0 = ok); no exceptions cross the ABI;
 *     afe_last_error() gives the message of the last failure on that engine.
 *   - host arrays are caller-allocated and PLANAR: a 3-vector field of `count`
 *     vehicles is x[0..count) y[0..count) z[0..count); quaternions are scalar
 *     first (w x y z), as Common/Common/Math/Rotation.hpp:46-51.
 *   - one engine = one ensemble on one GPU with one clock; one host thread at
 *     a time per engine (the reference is single-threaded as well).
 *   - stepping is asynchronous on the engine's HIP stream; getters synchronise.
 *   - there is NO CPU fallback: without a usable gfx950 device afe_create
 *     fails with AFE_ERR_NO_DEVICE.
 */
#ifndef AGRIFLY_ENGINE_H
#define AGRIFLY_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: afe_device_view starts with `struct_bytes` (set by the caller; the engine never writes past it) and carries
 *    pos_anchor_xy -- AFE_F32 engines keep x and y in `pos` RELATIVE to the last set point (round 3 changed the meaning of
 *    view.pos without changing this number: a host built against version 1 must be rebuilt, afe_abi_version() tells). */
/* 3: afe_set_reserved_compute_units is gone (no consumer; a reservation cost the grid 4-25 %); added
 *    afe_has_dev_hooks and afe_persistent_kernarg_layout.  Nothing else moved: a version-2 host that never called the
 *    removed function runs unchanged after a rebuild. */
#define AFE_ABI_VERSION 3

typedef struct afe_engine afe_engine; /* opaque */

typedef enum afe_status {
  AFE_OK = 0,
  AFE_ERR_INVALID_ARG = 1,
  AFE_ERR_NO_DEVICE = 2,   /* no HIP device / not gfx950 / kernels missing */
  AFE_ERR_HIP = 3,         /* a HIP runtime call failed (see afe_last_error) */
  AFE_ERR_OUT_OF_RANGE = 4,
  AFE_ERR_NOT_CONFIGURED = 5, /* stepping before a type table was set */
  AFE_ERR_COMM = 6
} afe_status;

typedef enum afe_precision {
  AFE_F32 = 0, /* fp32 state + arithmetic: the production path (176 B/step) */
  AFE_F64 = 1  /* fp64 state + arithmetic: the reference's own precision */
} afe_precision;

typedef enum afe_seed_policy {
  /* every vehicle seeds std::default_random_engine with 1, exactly like the
   * reference ensemble does (Quadcopter_T.cpp:27) */
  AFE_SEED_REFERENCE = 0,
  /* seed = 1 + global vehicle index: independent Monte-Carlo streams */
  AFE_SEED_DECORRELATED = 1,
  /* Monte-Carlo ensembles that need no libstdc++ stream: the six normals of (vehicle, logic tick) come from a
   * counter-based generator -- Philox4x32-10 keyed by afe_set_noise_seed, addressed by the GLOBAL vehicle index and
   * the tick number, Box-Muller on top:
   *   key = (seed low, seed high); counter = (index low, index high (16 bits) | stream << 16 | block << 24, ordinal low,
   *   ordinal high); stream 1 = IMU noise (ordinal = logic-tick number; blocks 0 and 1 give z0..z7: gyro x y z = z0 z1 z2,
   *   accelerometer x y z = z3 z4 z5), stream 2 = gusts (ordinal = epoch; block 0: force = sigma_i (z0, z1, z2));
   *   per pair of words: u_r = ((x_even >> 9) + 0.5) 2^-23, u_a = (x_odd >> 8) 2^-24 (both exact in fp32),
   *   r = sqrt(-2 ln u_r), z_a = r cos(2 pi u_a), z_b = r sin(2 pi u_a)   (|z| <= 5.65).
   * The AFE_F64 engine evaluates this in double (1e-13 of a libm evaluation), the AFE_F32 engine in float with the
   * hardware reciprocal / square root / sine / cosine (measured worst 6e-7 absolute).  No per-vehicle engine word
   * is loaded or stored, nothing diverges (no rejection loop); the samples do not depend on how the ensemble is
   * sharded, stepped or fused.  The reference produces no such stream (it seeds every vehicle with 1). */
  AFE_SEED_COUNTER = 2
} afe_seed_policy;

/* The per-vehicle constant record: the arguments of
 *   Simulation::Quadcopter_T<logicType>::Quadcopter_T(...)
 *   Components/Components/Simulation/Quadcopter_T.hpp:24-32
 * (masterTimer, id and quadcopterType are engine / facade concerns).  The
 * engine derives motor geometry, I^-1 and the IMU mount matrix from it the way
 * the ctor body does (Quadcopter_T.cpp:20,45-65,75-80). */
typedef struct afe_vehicle_params {
  double mass;                       /* [kg] */
  double inertia[9];                 /* row-major 3x3 [kg m^2] */
  double arm_length;                 /* [m] */
  double com_error[3];               /* centreOfMassError [m] */
  double motor_min_speed;            /* [rad/s] */
  double motor_max_speed;            /* [rad/s] */
  double prop_thrust_from_speed_sqr; /* [N/(rad/s)^2] */
  double prop_torque_from_speed_sqr; /* [N m/(rad/s)^2] */
  double motor_time_const;           /* [s]; 0 = instantaneous */
  double motor_inertia;              /* [kg m^2] */
  double lin_drag_coeff_b[3];        /* [N s/m], body axes */
  float imu_yaw, imu_pitch, imu_roll; /* QuadcopterConstants::IMU_* [rad] */
} afe_vehicle_params;

/* ---- vehicle-type table -------------------------------------------------
 * Replaces Onboard::QuadcopterConstants(QuadcopterType) +
 * GetVehicleTypeFromID (Components/Components/Logic/QuadcopterConstants.hpp:
 * 31-274,297-332) as consumed by Simulator/Rappids_Simulator/main.cpp:147-218.
 * type: 1 CF_STANDARD, 2 CF_BIGMOTORSPROPS, 4 CF_LARGEQUAD, 5 CF_MINIQUAD.
 * Pure host functions: usable without a GPU. */
int afe_params_from_type(int quadcopter_type, afe_vehicle_params *out);
int afe_type_from_id(unsigned vehicle_id);

/* ---- lifetime -----------------------------------------------------------
 * Replaces `new Simulation::Quadcopter(...)` x n_vehicles (main.cpp:211-218;
 * the multi-vehicle form is the std::vector<SimVehicle> of
 * AIFS_ROS/hiperlab_rostools/src/Simulator/main.cpp:58-95).
 * device < 0 selects the current HIP device.  first_global_index is this
 * shard's offset in the whole ensemble (used by AFE_SEED_DECORRELATED and by
 * afe_nearest_neighbour's self-exclusion); 0 for a single-GPU ensemble.  Initial state is the
 * reference's: origin, at rest, identity attitude, motors stopped, engine
 * clock 0 (SimulationObject6DOF.hpp:14-19, Motor.cpp:18). */
int afe_create(afe_engine **out, int64_t n_vehicles, int precision, int device,
               int64_t first_global_index);
/* The same engine with its state arena in pinned, coherent HOST memory that the device addresses over the bus -- for
 * small ensembles with the host in the loop of every step (one `Quadcopter_T::Run()` + `GetPosition()` per millisecond,
 * Simulator/Rappids_Simulator/main.cpp:330-392; the onboard logic of Quadcopter_T.cpp:159-189 on the host).  Getters
 * and setters of state, commands, wrench and IMU are then plain copies inside host memory once the steps authorised so
 * far are done: no transfer call, and a resident grid (the engine starts in AFE_STEP_AUTO) is NOT parked by them.
 * One step of that loop then costs ~9 us instead of ~75 (one vehicle, MI355X).  A step's loads and stores cross the bus
 * (~1.5 us of latency, all in flight together), so this is the wrong
 * choice for ensembles that step many times between host visits or hold more than a few thousand vehicles.
 * Same results, bit for bit, as afe_create; every other entry point behaves the same. */
int afe_create_host_visible(afe_engine **out, int64_t n_vehicles, int precision, int device,
                            int64_t first_global_index);
int afe_destroy(afe_engine *e);
const char *afe_last_error(const afe_engine *e);
const char *afe_status_string(int status);
int afe_abi_version(void);
/* 1 when the loaded library was built with -DAFE_DEV_HOOKS (the kernel lab's measurement variables of tools/ are
 * read), 0 for the release build, which reads exactly six environment variables:
 * AFE_PERSIST_AQL, AFE_FORCE_STEP_MODE, AFE_FORCE_SPLIT, AFE_FORCE_HOST_ARENA, AFE_PERSIST_DEBUG, AFE_GRID_LOG. */
int afe_has_dev_hooks(void);
/* Build self-check (no GPU needed): byte offset and size of each of the four by-value arguments of the resident step
 * kernel -- StepView, DevParams, DevLogic, PersistArgs -- as the HOST packs them for a dispatch on the engine's own
 * queue (precision AFE_F32 / AFE_F64), and the size of the whole segment.  tests/test_kernel_resources.py holds them
 * against the argument table of every instantiation in the gfx950 code object inside the library. */
int afe_persistent_kernarg_layout(int precision, int32_t offsets[4], int32_t sizes[4], int32_t *segment_bytes);

/* Use a caller-owned HIP stream (hipStream_t) for all engine work, e.g.
 * torch's current stream.  NULL restores the engine's own stream. */
int afe_set_stream(afe_engine *e, void *hip_stream);

/* ---- configuration ------------------------------------------------------ */
/* Table of up to 256 parameter records; vehicle i uses table[type_index[i]].
 * Replaces the per-object ctor args.  How the step kernel gets at a record:
 * all vehicles on record 0 -> kernel arguments; the type index constant over
 * every aligned run of 64 vehicles (a fleet laid out type by type) -> one
 * scalar-loaded record per wave, as fast as the homogeneous case whatever the
 * table size; types mixed within a run of 64 -> the table is staged into LDS
 * per workgroup (slower as the table grows).  Results are the same bits. */
int afe_set_type_table(afe_engine *e, const afe_vehicle_params *table, int n_types);
int afe_set_vehicle_types(afe_engine *e, int64_t first, int64_t count,
                          const uint8_t *type_index);
/* onboardLogicPeriod ctor argument (Quadcopter_T.hpp:32, used at
 * Quadcopter_T.cpp:159-160).  Default 1/500 s (main.cpp:177). */
int afe_set_logic_period(afe_engine *e, double seconds);
/* IMU noise (Quadcopter_T.cpp:5-6,165-180).  The reference hard-codes
 * sigma_gyro = 0.1, sigma_acc = 0.2 and seed 1; these are the defaults.
 * enabled = 0 switches the six draws off (the stream does not advance).
 * The std::minstd_rand0 words and the polar method's accept / reject decisions
 * are libstdc++'s bit for bit in both precisions; the six normal values are
 * libstdc++'s doubles (to 4e-15) in the AFE_F64 engine and float evaluations of
 * the same expression (~4e-7 relative) in the AFE_F32 engine. */
int afe_set_imu_noise(afe_engine *e, int enabled, double sigma_gyro,
                      double sigma_acc, int seed_policy);

/* ---- state in / out -----------------------------------------------------
 * Replace SimulationObject6DOF::Set/Get{Position,Velocity,Attitude,
 * AngularVelocity} (Components/Components/Simulation/SimulationObject6DOF.hpp:
 * 26-56) and Motor::SetSpeed / the motor speed read behind GetMotorForce
 * (Motor.hpp:30-32, Quadcopter_T.hpp:39).  NULL field pointers are skipped. */
int afe_set_state(afe_engine *e, int64_t first, int64_t count,
                  const double *pos3, const double *vel3, const double *att4,
                  const double *ang_vel3, const double *motor_speed4);
int afe_get_state(afe_engine *e, int64_t first, int64_t count, double *pos3,
                  double *vel3, double *att4, double *ang_vel3,
                  double *motor_speed4);
int afe_set_state_f32(afe_engine *e, int64_t first, int64_t count,
                      const float *pos3, const float *vel3, const float *att4,
                      const float *ang_vel3, const float *motor_speed4);
int afe_get_state_f32(afe_engine *e, int64_t first, int64_t count, float *pos3,
                      float *vel3, float *att4, float *ang_vel3,
                      float *motor_speed4);
/* per-vehicle std::minstd_rand0 word (Quadcopter_T.hpp:122); checkpointing */
int afe_set_rng_state(afe_engine *e, int64_t first, int64_t count, const uint32_t *state);
int afe_get_rng_state(afe_engine *e, int64_t first, int64_t count, uint32_t *state);

/* ---- per-tick inputs ----------------------------------------------------
 * afe_set_motor_cmds replaces the read-back of logicType::GetMotorSpeedCmd(i)
 * into _motorSpeedCommands (Quadcopter_T.cpp:187-189; float, Quadcopter_T.hpp:
 * 100).  afe_set_external_force / _torque replace SetExternalForce /
 * SetExternalTorque (Quadcopter_T.hpp:45-51; world frame, persist until
 * replaced).  Passing NULL to the external setters zeroes the range; the
 * kernel only reads the wrench arrays once one has been set non-NULL. */
int afe_set_motor_cmds(afe_engine *e, int64_t first, int64_t count, const float *cmd4);
int afe_set_external_force(afe_engine *e, int64_t first, int64_t count, const double *force3);
int afe_set_external_torque(afe_engine *e, int64_t first, int64_t count, const double *torque3);

/* ---- on-device onboard rates logic (optional) -------
 * Instead of handing each IMU sample to a host-side logicType, the step kernel
 * can run the rates-control slice of Onboard::QuadcopterLogic itself at every
 * logic tick, closing the loop on the GPU:
 *   SetIMUMeasurementRateGyro: _R * gyro -> 2nd-order low-pass
 *       (Components/Components/Logic/QuadcopterLogic.hpp:40-45,
 *        Common/Common/Math/LowPassFilterSecondOrder.hpp:22-64)
 *   KalmanFilter6DOF::Predict without UWB: angular-velocity estimate = filtered
 *       gyro; the first call only initialises (KalmanFilter6DOF.cpp:70-116)
 *   RunControllerExternalRatesControl (QuadcopterLogic.cpp:528-541):
 *       QuadcopterAngularVelocityController::GetDesiredTorques
 *       (QuadcopterAngularVelocityController.hpp:25-38),
 *       QuadcopterMixer::GetMotorForces / PropellerSpeedsFromThrust
 *       (QuadcopterMixer.hpp:63-99)
 * and write the four motor-speed commands, which act from the next step on as
 * in Quadcopter_T.cpp:187-189.  Not included: the flight-state machine beyond
 * IDLE -> EXTERNAL_RATES_CONTROL, panic checks, attitude estimate, telemetry,
 * motor calibration (host-side logic).  All logic arithmetic is float, as in
 * the reference.  Record i of the logic table pairs with record i of the
 * vehicle type table. */
typedef struct afe_rates_logic_params {
  float mass;                       /* QuadcopterLogic::_mass */
  float inertia[9];                 /* QuadcopterConstants::inertiaMatrix */
  float ang_vel_time_const_xy;      /* angVelControl_timeConst_xy [s] */
  float ang_vel_time_const_z;       /* angVelControl_timeConst_z [s] */
  float arm_length;                 /* [m] */
  float prop_thrust_from_speed_sqr; /* [N/(rad/s)^2] */
  float prop_torque_from_thrust;    /* [N m/N] */
  int prop0_spin_dir;               /* +1 / -1 */
  float max_thrust_per_propeller;   /* [N] */
  float min_thrust_per_propeller;   /* [N] */
  float max_cmd_total_thrust;       /* [N]; < 0: mixer default 4*max*0.8 */
  float imu_yaw, imu_pitch, imu_roll; /* [rad] */
  float gyro_lowpass_cutoff;        /* [rad/s]; the reference uses 200 */
} afe_rates_logic_params;
/* QuadcopterConstants(type) narrowed to the fields above. Pure host. */
int afe_rates_logic_params_from_type(int quadcopter_type, afe_rates_logic_params *out);
/* Enable (table != NULL) or disable (table == NULL) the on-device logic.
 * Enabling resets the logic state: filters at 0, estimator uninitialised,
 * flight state IDLE (motor commands 0) until afe_set_rates_commands.  The
 * filter sampling period is float(logic period), as Quadcopter_T.cpp:18. */
int afe_set_rates_logic(afe_engine *e, const afe_rates_logic_params *table, int n_types);
/* The decoded externalRatesCmd radio message (RadioTypes.hpp:218-226):
 * floats[0] = total thrust normalised by mass [m/s^2], floats[1..3] = desired
 * body rates [rad/s].  Persists until replaced; puts the vehicles of the range
 * into EXTERNAL_RATES_CONTROL. */
int afe_set_rates_commands(afe_engine *e, int64_t first, int64_t count,
                           const float *thrust_norm, const float *ang_vel3);
/* The motor-speed commands currently in force (host- or logic-written). */
int afe_get_motor_cmds(afe_engine *e, int64_t first, int64_t count, float *cmd4);

/* ---- wire formats either side of the step (host-only) --
 * Byte-exact equivalents of RadioTypes::RadioMessageDecoded (Common/Common/
 * DataTypes/RadioTypes.hpp:39-246: 23-byte uplink, type@0 reserved@1 flags@2,
 * ten big-endian 16-bit fixed-point fields) and of TelemetryPacket::
 * Encode/DecodeTelemetryPacket (TelemetryPacket.hpp:32-207: packed
 * {u8 type; u8 packetNumber; u16 data[14]} = 30 bytes).  Usable without a GPU. */
#define AFE_RADIO_PACKET_SIZE 23
#define AFE_TELEMETRY_PACKET_SIZE 30
typedef struct afe_radio_message {  /* RadioMessageDecoded: type, flags, floats[10] */
  uint8_t type;   /* 2 kill, 3 position, 4 acceleration, 5 rates, 6 idle (RadioTypes.hpp:17-25) */
  uint8_t flags;
  float floats[10];
} afe_radio_message;
int afe_radio_create_rates_command(uint8_t flags, float des_total_thrust, const float des_ang_vel[3],
                                   uint8_t raw_out[AFE_RADIO_PACKET_SIZE]);          /* :158-171 */
int afe_radio_create_position_command(uint8_t flags, const float pos[3], const float vel[3],
                                      const float acc[3], uint8_t raw_out[AFE_RADIO_PACKET_SIZE]); /* :137-156 */
int afe_radio_create_acceleration_command(uint8_t flags, const float acc[3], float yaw_rate,
                                          uint8_t raw_out[AFE_RADIO_PACKET_SIZE]);   /* :173-187 */
int afe_radio_create_simple_command(int type /* 2 kill | 6 idle */, uint8_t flags,
                                    uint8_t raw_out[AFE_RADIO_PACKET_SIZE]);         /* :123-135 */
int afe_radio_decode(const uint8_t raw[AFE_RADIO_PACKET_SIZE], afe_radio_message *out); /* :189-240 */
typedef struct afe_telemetry_packet {  /* TelemetryPacket::TelemetryPacket, TelemetryPacket.hpp:102-120 */
  uint8_t type;            /* 0 = part 1 (accel gyro motorForces position battVoltage), 1 = part 2 */
  uint8_t packet_number;
  float accel[3], gyro[3], motor_forces[4], position[3], batt_voltage;
  float velocity[3], attitude[3], debug_vals[6];
  uint8_t panic_reason, warnings;
} afe_telemetry_packet;
int afe_telemetry_encode(const afe_telemetry_packet *src, uint8_t out[AFE_TELEMETRY_PACKET_SIZE]);
int afe_telemetry_decode(const uint8_t in[AFE_TELEMETRY_PACKET_SIZE], afe_telemetry_packet *out);
/* SetCommandRadioMsg (Quadcopter_T.hpp:62-67) for the on-device logic: decode
 * `count` consecutive 23-byte packets and apply them to vehicles first..: rates
 * commands enter EXTERNAL_RATES_CONTROL, idle / kill return to zero motor
 * commands; other types are refused with AFE_ERR_INVALID_ARG (those flight
 * modes live in the host-side logic). */
int afe_set_commands_from_radio(afe_engine *e, int64_t first, int64_t count, const uint8_t *raw_packets);

/* ---- batched RAPPIDS depth-image planner --------------
 * The caller on the other side of the step: per vehicle, search random minimum-
 * jerk motion primitives for the lowest-cost one that is input-feasible,
 * velocity-admissible and collision-free against a depth image, exactly as
 * RectangularPyramidPlanner::DepthImagePlanner::FindLowestCostTrajectory
 * (Components/Components/DepthImagePlanner/DepthImagePlanner.cpp:91-212) with
 * RandomTrajectoryGenerator (DepthImagePlanner.hpp:335-432), IsCollisionFree
 * (:214-301), InflatePyramid (:456-970) and RapidTrajectoryGenerator
 * (Components/Components/TrajectoryGenerator/RapidTrajectoryGenerator.cpp:
 * 75-208), with ONE deliberate change: the reference stops after a wall-clock
 * budget (main.cpp:498: 50 ms), which makes the number of candidates
 * nondeterministic; here it is an explicit count.  Everything is double, in the
 * camera-fixed frame (initial position 0; x right, y down, z into the image). */
typedef struct afe_planner_config {
  int width, height;                 /* depth image size [pixels] */
  double depth_scale;                /* metres per count (main.cpp:121-122: 10/256) */
  double focal_length, cx, cy;       /* pinhole intrinsics [pixels] (main.cpp:360,484-488) */
  double true_vehicle_radius;        /* physicalVehicleRadius [m] */
  double planning_vehicle_radius;    /* vehicleRadiusForPlanning [m] */
  double min_checking_dist;          /* minimumCollisionDistance [m] */
  double min_thrust, max_thrust;     /* [m/s^2]; reference defaults 5, 30 (DepthImagePlanner.cpp:43-44) */
  double max_ang_vel;                /* [rad/s]; default 20 */
  double max_velocity;               /* [m/s]; default 5 */
  double min_section_time;           /* [s]; default 0.02 */
  int max_pyramids;                  /* per plan (SetMaxNumberOfPyramids); also sizes the scratch */
  int pixel_buffer;                  /* _pyramidSearchPixelBuffer = 2 */
  int cost_type;                     /* 0: ExplorationCost, -dir.pos(T)/T (DepthImagePlanner.hpp:476-506)
                                        1: Rappids_Simulator's goal cost, -(|G|-|G-pos(T)|)/T (main.cpp:86-107) */
  double cost_vec[3];                /* direction or goal (camera frame) when no per-vehicle array is given */
} afe_planner_config;
int afe_planner_default_config(afe_planner_config *out, int width, int height, double depth_scale,
                               double focal_length, double true_vehicle_radius,
                               double planning_vehicle_radius, double min_checking_dist);
typedef struct afe_plan_output {
  int found;            /* FindLowestCostTrajectory's return value */
  int best_index;       /* winning candidate, -1 if none */
  double best_cost;
  double coeffs[6][3];  /* CommonMath::Trajectory of the winner: t^5 .. t^0 (Trajectory.hpp:31-36) */
  double tf;            /* its duration [s] */
  int n_generated, n_cost_checks, n_collision_checks, n_velocity_checks, n_collision_free, n_pyramids;
} afe_plan_output;
/* Candidate end states exactly as RandomTrajectoryGenerator's default constructor
 * draws them from std::mt19937(seed) (DepthImagePlanner.hpp:349-366,393-404, with
 * GCC's right-to-left evaluation of the three arguments): samples[k] = {pixelX,
 * pixelY, depth [m], duration [s]}.  The reference re-seeds with 0 at every call.
 * Pure host. */
int afe_planner_samples(uint32_t seed, int width, int height, int n_candidates, double *samples4);
/* Plan for n vehicles on GPU `device` (< 0: current).  Host arrays:
 *   depth_images [n_images][height][width] uint16; image_index[n] or NULL (image i)
 *   vel0, acc0, grav: planar [3][n]; cost_vec planar [3][n] or NULL (cfg->cost_vec)
 *   samples [n_tables][n_candidates][4]; sample_table[n] or NULL (table 0)
 *   out[n]; flags [n][n_candidates] or NULL: TrajectoryTestResult bits per candidate
 *   (1 LowCost, 2 DynamicsFeasible, 4 VelocityAdmissible, 8 CollisionFree).
 * Image size: ceil(width/64) * height <= 8192 (one bit per pixel is kept in LDS;
 * e.g. 640x480, 1024x512), else AFE_ERR_OUT_OF_RANGE. */
int afe_rappids_plan(int device, const afe_planner_config *cfg, int64_t n, const uint16_t *depth_images,
                     int64_t n_images, const int32_t *image_index, const double *vel0, const double *acc0,
                     const double *grav, const double *cost_vec, const double *samples, int n_tables,
                     const int32_t *sample_table, int n_candidates, afe_plan_output *out, uint8_t *flags,
                     float *kernel_ms);

/* ---- depth camera --------------------------------------
 * The reference gets its depth image from AirSim/Unity over RPC
 * (Simulator/Rappids_Simulator/main.cpp:332-354: ImageType::DepthVis, 8-bit,
 * widened to uint16; far = 10 m, depthScale = far/256, focal = width/2,
 * :120-122,360) with the camera mounted at depthCamAtt = FromEulerYPR(-90 deg,
 * 0, -90 deg) on the body (:123-125; camera-to-world = att * depthCamAtt,
 * :520).  Neither the renderer nor the orchard scene is in the reference tree,
 * so what is replaced here is that RPC and its image contract: one ray per
 * pixel through ((x - cx)/f, (y - cy)/f, 1) in the camera frame (x right,
 * y down, z forward), closest hit over a triangle mesh, count =
 * min(max_count, floor(z / depth_scale)), max_count for a miss.
 *
 * A scene is a static triangle mesh (world frame, metres) with a bounding-
 * volume hierarchy built on the host and kept in HBM. */
typedef struct afe_scene afe_scene;

typedef struct afe_camera {
  int32_t width, height;
  double focal_length, cx, cy;
  double depth_scale;   /* metres per count */
  int32_t max_count;    /* 255: 8-bit DepthVis */
  int32_t reserved;
} afe_camera;

/* main.cpp's camera: focal = width/2, principal point = centre, 10 m / 256. */
int afe_camera_default(afe_camera *cam, int width, int height);
/* depthCamAtt of main.cpp:123-125 as a quaternion (w, x, y, z). */
int afe_camera_default_mount(double mount[4]);

/* triangles: n_tri x 9 floats (v0, v1, v2).  device < 0: current device. */
int afe_scene_create(int device, const float *triangles, int64_t n_tri, afe_scene **out);
void afe_scene_destroy(afe_scene *s);
/* Pure host (no GPU): builds the hierarchy afe_scene_create would build and verifies it --
 * every triangle in exactly one leaf, every box containing what hangs below it, depth within
 * the traversal stack; returns its node count, depth and largest leaf. */
int afe_scene_check_hierarchy(const float *triangles, int64_t n_tri, int64_t *n_nodes, int *depth, int *max_leaf);
/* n_tri, number of BVH nodes, tree depth, world bounds {min xyz, max xyz} */
int afe_scene_info(const afe_scene *s, int64_t *n_tri, int64_t *n_nodes, int *depth, double bounds[6]);

/* n_views depth images from explicit poses.  pos: planar [3][n_views]
 * doubles, att: planar [4][n_views] (w,x,y,z) body attitudes, mount: body-to-
 * camera mount quaternion applied to every view (NULL = identity, att is then
 * the camera attitude itself).  depth_out: host buffer of n_views * height *
 * width uint16.  kernel_ms (optional): HIP-event time of the render launch. */
int afe_render_depth(afe_scene *s, const afe_camera *cam, int64_t n_views, const double *pos, const double *att,
                     const double mount[4], uint16_t *depth_out, float *kernel_ms);

/* Which form of the traversal the renders of this scene use: 0 (default) -- tiles whose 64 rays agree on
 * the direction signs walk the octant-mirrored copy of the tree (ordered box tests), the others the plain
 * copy; 1 -- every tile takes the plain, sign-agnostic walk.  The images are the same bits either way
 * (tests/test_gpu_render.py renders thousands of views both ways): a cross-check, not a tuning knob. */
int afe_scene_set_walk(afe_scene *s, int mode);

/* What the traversal did for such a batch (a counting build of the same kernel; the images are
 * discarded): stats[0] BVH nodes visited and [1] triangle box tests / [2] double-precision
 * ray-triangle tests executed, per wave of 64 rays; [3], [4] the same two per participating ray;
 * [5] rays; [6] waves; [7] the triangles the tiles really show (distinct closest-hit triangles per wave, summed):
 * the double-precision tests no traversal could avoid.  bench.py states the camera's floor with these. */
int afe_render_depth_stats(afe_scene *s, const afe_camera *cam, int64_t n_views, const double *pos, const double *att,
                           const double mount[4], uint64_t stats[8], float *kernel_ms);

/* The same, with the poses read on the device from the engine's state slabs
 * (vehicles [first, first+count)), replacing client.simGetImages() for every
 * vehicle at once.  depth_out is a DEVICE pointer when out_is_device != 0
 * (count * height * width uint16, e.g. the buffer handed to
 * afe_rappids_plan_device), else a host buffer.  Engine and scene must live on
 * the same device; the launch is ordered on the engine's stream. */
int afe_render_depth_engine(afe_engine *e, afe_scene *s, const afe_camera *cam, int64_t first, int64_t count,
                            const double mount[4], void *depth_out, int out_is_device, float *kernel_ms);

/* Device scratch helpers for hosts without their own HIP allocator (ctypes). */
int afe_device_alloc(int device, uint64_t bytes, void **out);
int afe_device_free(void *p);
int afe_device_download(void *host_dst, const void *dev_src, uint64_t bytes);

/* The planner keeps its device scratch between calls (grown on demand, one set per process; plan calls take
 * turns on it).  This gives the memory back; the next call allocates again. */
int afe_planner_release_scratch(void);

/* afe_rappids_plan with the depth images already in HBM (one per planner, or
 * indexed through image_index, which stays a host array).  The pointer must be
 * 16-byte aligned (anything from afe_device_alloc / hipMalloc is). */
int afe_rappids_plan_device(int device, const afe_planner_config *cfg, int64_t n, const void *dev_depth_images,
                            int64_t n_images, const int32_t *image_index, const double *vel0, const double *acc0,
                            const double *grav, const double *cost_vec, const double *samples, int n_tables,
                            const int32_t *sample_table, int n_candidates, afe_plan_output *out, uint8_t *flags,
                            float *kernel_ms);

/* ---- stepping -----------------------------------------------------------
 * afe_step replaces the loop body
 *     for (v : vehicles) v->Run();  simTimer.AdvanceMicroSeconds(dt_us);
 * (Simulator/Rappids_Simulator/main.cpp:391-392; AIFS_ROS/.../Simulator/
 * main.cpp:323-325) n_steps times: every vehicle takes n_steps physics steps
 * of dt = dt_us * 1e-6 s (Quadcopter_T.cpp:85-156, Motor.cpp:39-84).  The
 * engine clock advances by dt_us per step; whenever the onboard-logic gate of
 * Quadcopter_T.cpp:159-160 fires (strict >, then minus one period) the step
 * also synthesises the IMU sample of :165-180 into the IMU buffers and counts
 * one logic tick.  dt_us == 0 is the reference's "dt < 1e-6: return".
 * Within one call the motor commands and external wrench are held constant,
 * as they are between two logicType::Run() calls in the reference.
 * afe_steps_until_tick tells a host-side logicType driver how many steps of
 * dt_us may be fused before its Run() is due (>= 1). */
int afe_step(afe_engine *e, uint64_t dt_us, int n_steps);
/* How the step kernels address the state slabs: 0 (default) -- through buffer resources spanning the engine's
 * arenas whenever those fit 32-bit offsets (up to ~31 M fp32 vehicles), otherwise by global addresses; 1 -- always by
 * global addresses.  Same results bit for bit; mode 1 exists so that the kernels very large ensembles run are
 * testable at any size. */
int afe_set_addressing(afe_engine *e, int mode);
/* Which step kernel the next afe_step will launch: record_path 0 = parameters in the kernel arguments (all
 * vehicles on record 0), 1 = one scalar-loaded record per wave (type constant over every aligned run of 64),
 * 2 = type table in LDS; addressing 0 = buffer resources, 1 = global addresses.  Either pointer may be NULL. */
int afe_step_kernel_info(const afe_engine *e, int *record_path, int *addressing);

/* Split stepping.  With two parts afe_step launches the first half of the ensemble on the engine's stream and the
 * second half on a stream of its own, and the two chains of launches never wait for each other -- each one's
 * drain-and-dispatch gap (a fixed ~2.7 us per launch) is covered by the other's streaming.  At 2^20 vehicles a
 * 1 ms step takes 22.1 instead of 24.4 us; per-vehicle results are the same bits (vehicles do not interact).
 * What changes is WHEN the engine's stream is ordered after the steps: not at the return of afe_step but at the
 * next engine call that touches device state or the stream (afe_sync, getters and setters, afe_event_record,
 * afe_pack_positions, the queries, the depth camera, checkpoints, ... every entry point but afe_step joins the
 * two streams first).  While the engine uses its OWN stream nobody else can queue work on it, so the difference
 * cannot be observed.
 *   parts = 0 (default): automatic -- two parts for ensembles of 2^19 vehicles and more as long as the engine owns
 *     its stream; one launch on one stream otherwise (smaller ensembles lose: the second launch costs more host
 *     time than it hides; a caller-owned stream keeps the order-at-return behaviour).
 *     (Engines an afe_group creates for several devices start with parts = 1: one host thread launches for all of
 *     them, and a second launch per shard and step would make that thread the limit.)
 *   parts = 1: never.   parts = 2: always (ensembles of 1 024 vehicles and more), also on a caller-owned stream
 *     (afe_set_stream) -- a host that queues its own work there and expects it to see the stepped state must
 *     then call afe_sync or afe_event_record in between. */
int afe_set_split_stepping(afe_engine *e, int parts);

/* Key of the counter-based noise generator (AFE_SEED_COUNTER); default 0. */
int afe_set_noise_seed(afe_engine *e, uint64_t seed);

/* BASELINE config 4's disturbance process on the device: a per-vehicle wind-gust force through the SetExternalForce
 * port (Components/Components/Simulation/Quadcopter_T.hpp:45, applied at Quadcopter_T.cpp:132; the reference has the
 * port and no model behind it).  While enabled the engine owns the external-force slab: during epoch
 * k = floor(step start time / period_us) vehicle i feels F = sigma_i (z0, z1, z2) with sigma_i = sigma_max * g / (n_global - 1),
 * g = first_global_index + i, and z = N(0,1) from the counter-based generator at (seed, g, epoch k) -- piecewise
 * constant, resampled on the device when a step starts in a new epoch ("N(0, sigma^2), sigma swept
 * 0...0.5 N, piecewise-constant 100 ms").  No host upload, no per-step cost; the force of (vehicle, epoch) does not
 * depend on sharding or on how the steps are issued.  n_global <= 0: this ensemble alone.  enabled = 0 leaves the
 * slab as it is (afe_set_external_force takes over).  afe_get_external_force reads the slab back. */
int afe_set_gust_process(afe_engine *e, int enabled, uint64_t seed, double sigma_max, uint64_t period_us, int64_t n_global);
int afe_get_external_force(afe_engine *e, int64_t first, int64_t count, double *force3);

/* Persistent stepping: afe_step without a kernel launch per step.  The step loop of the reference
 * (`for each vehicle: Run(); clock += dt`, Simulator/Rappids_Simulator/main.cpp:330,391-392;
 * AIFS_ROS/hiperlab_rostools/src/Simulator/main.cpp:323-325) has no barrier between vehicles, so nothing in it
 * asks for one between two steps of different vehicles either.  In this mode ONE resident grid of one-wave
 * workgroups stays on the device; afe_step(e, dt, k) authorises k more steps by writing k 8-byte descriptors
 * (step number, logic-tick flag) into a ring in pinned host memory and returns; every wave advances ITS vehicles
 * through each descriptor as it appears -- loads, the step, stores: the body of the one-step launch, the state
 * back in HBM after every step, bit for bit the launched kernels' results.  What a dependent launch costs on top
 * of its streaming time (2.7 us here: drain, dispatch, ramp-up) is not paid; an ensemble of 131 072 vehicles --
 * one GPU's shard of the 1 M-vehicle configuration on eight -- steps in [see DESIGN.md section 6] instead of 5.6 us.
 * The grid leaves the device ("parks") when any other entry point needs the stream or the state (afe_sync, getters,
 * setters, events, queries, checkpoints ...: they all do it implicitly and the next afe_step starts a new grid), and
 * by itself when the host has not authorised a step for ~200 us, so nothing -- not even a device-wide
 * synchronisation issued elsewhere -- can wait on it for ever.  A grid of a large ensemble occupies every wave slot of
 * the device while it is resident: another engine's grid, or any other kernel of the process, starts when it has left
 * (two such engines stepped alternately take turns, each handing over after its 200 us of idling -- correct, slow:
 * tests/test_gpu_persistent.py; one process, one large resident grid per device is the intended use).  A grid that starts
 * while another engine's launch is running may find the register file fragmented and a few of its workgroups without a
 * slot; the grid notices (60 ms), parks at one step and is started again -- results are unaffected, stderr says so.
 *   mode = AFE_STEP_LAUNCH (0, default): one kernel launch per step (or per afe_set_max_fused_steps chunk).
 *   mode = AFE_STEP_PERSISTENT (1): as above, whenever the ensemble qualifies -- every vehicle on type record 0,
 *     no external torque, the engine's own stream, an arena below 4 GiB; otherwise the launches, silently.
 *   mode = AFE_STEP_AUTO (2): the engine picks per call.  One step per call: the resident grid in its resident-state
 *     form (AFE_STEP_RESIDENT below: the same bits, fewer bytes) for ensembles of up to 2^20 vehicles, launches (split,
 *     see above; cache-policy hints by size, afe_set_cache_policy) beyond.  Several steps per call (nothing is observable in between): from
 *     8 steps, or from 2 at 2^19 vehicles and more, fused launches -- the state stays in registers from step to step --
 *     which is the fastest way to the same bits there (measured table: DESIGN.md section 6).
 * afe_steps_completed: the number of steps (since afe_create) that EVERY vehicle has been advanced through -- the
 * per-step completion word of the resident grid, readable at any time without disturbing it; in launch mode the
 * number of steps issued.  (Reading the state itself goes through the getters, which park the grid first.) */
#define AFE_STEP_LAUNCH 0
#define AFE_STEP_PERSISTENT 1
#define AFE_STEP_AUTO 2
/*   mode = AFE_STEP_RESIDENT (3): the resident grid with the steps that are ALREADY authorised taken together: a wave
 *     that finds k steps waiting loads its vehicles' inputs once, keeps the state in registers from step to step and
 *     stores each step's state as it is made -- every step still lands in memory (a host that authorises one step at a
 *     time and watches afe_steps_completed sees exactly the persistent mode), but a host that runs ahead pays for the
 *     stores only (~64 instead of ~143 bytes per vehicle-step).  Same bits.  Not what bench.py's headline measures
 *     (that one reads the state back every step); reported beside it. */
#define AFE_STEP_RESIDENT 3
int afe_set_step_mode(afe_engine *e, int mode);
int afe_steps_completed(afe_engine *e, uint64_t *steps);
/* 1 while a resident grid is on the device, 0 otherwise (diagnostic; tests use it) */
int afe_persistent_running(const afe_engine *e, int *running);
/* Device time spent by the resident grids since the last call (begin / end timestamps of each grid's dispatch on the
 * engine's own queue) and the steps they served: what rocprofv3 --kernel-trace reports as the kernel's duration,
 * measured in the run.  Ends the grid now resident (it is counted).  Both 0 when the grids ran on the HIP stream
 * (AFE_PERSIST_AQL=0, a caller's stream): use afe_event_record there.  Replaces nothing in the reference. */
int afe_grid_time(afe_engine *e, uint64_t *device_ns, uint64_t *steps);
/* Where the resident grid is dispatched: 1 = a user-mode queue of the engine's own (the grid survives afe_sync, no HIP
 * synchronisation of the process waits for it; a dispatch costs the host 14 us), 0 = the engine's HIP stream (every
 * afe_sync ends the grid, a launch costs 3 us, workers that are done leave), -1 = automatic (default): the own queue up to
 * 262 144 vehicles, where it is faster and the grid leaves a third of the wave slots to others; the HIP stream beyond
 * (at 524 288 the own queue is 5 % faster in 20-step blocks but the grid fills the device: DESIGN.md section 6).
 * Same bits either way.  Ends the grid now resident.  Replaces nothing in the reference. */
int afe_set_resident_queue(afe_engine *e, int mode);

/* Cache-policy hints of the one-step launches' slab accesses (`nt` bits on the buffer instructions; never a different
 * result bit).  -1 automatic (default): by what the 256 MiB Infinity Cache can keep from one step to the next --
 * everything (up to ~2^20 fp32 vehicles): default policy; only the state (up to ~4 M): inputs (commands, wrench) and
 * outputs (IMU samples) are streamed past it; not even the state: everything is streamed, one contiguous range per XCD.
 * 0 / 1 / 2 / 3 force one of these.  Replaces nothing in the reference (a CPU's caches decide for themselves). */
int afe_set_cache_policy(afe_engine *e, int policy);
/* the policy (0..3) the next one-step launch of this engine would carry: the forced one, or what the automatic rule
 * resolves to for the engine's size and configuration (diagnostic; tests use it) */
int afe_cache_policy_in_use(const afe_engine *e, int *policy);

/* How many sub-steps afe_step may fuse into one kernel launch (1..64, default
 * 64).  1 = one launch per step (state goes through HBM every step: the
 * per-step-observable mode bench.py reports); results are bitwise identical
 * for every setting. */
int afe_set_max_fused_steps(afe_engine *e, int max_steps_per_launch);
int afe_steps_until_tick(const afe_engine *e, uint64_t dt_us, int *n_steps);
int afe_sync(afe_engine *e);
int afe_time_us(const afe_engine *e, uint64_t *now_us);
int afe_logic_ticks(const afe_engine *e, uint64_t *n_ticks);

/* The IMU sample handed to logicType::SetIMUMeasurementRateGyro /
 * SetIMUMeasurementAccelerometer at the most recent logic tick
 * (Quadcopter_T.cpp:171,180); floats, like the reference's Vec3f. */
int afe_get_imu(afe_engine *e, int64_t first, int64_t count, float *gyro3, float *acc3);

/* ---- checkpoint / resume -------------------------------------------------
 * The reference has none (its motor speeds, timers, RNG and logic state have no
 * accessor).  Here the SoA slabs ARE the checkpoint: state,
 * motor speeds and commands, wrench, last IMU sample, RNG words, on-device logic
 * state and the engine clock.  afe_checkpoint_size gives the byte count for the
 * current configuration.  A checkpoint loads into any engine (the one that saved
 * it or a freshly created one) with the same n_vehicles / precision that has been
 * given the same type table and on-device logic table: both are fingerprinted in
 * the header and a mismatch is refused (AFE_ERR_INVALID_ARG).  The per-vehicle
 * type indices, noise switch, sigmas, seed policy, logic period, wrench flags and
 * the clock are restored from the checkpoint. */
int afe_checkpoint_size(const afe_engine *e, uint64_t *bytes);
int afe_save_checkpoint(afe_engine *e, void *host_buffer, uint64_t bytes);
int afe_load_checkpoint(afe_engine *e, const void *host_buffer, uint64_t bytes);

/* Pure host helper (no GPU): the logic-gate pattern of n_steps steps of dt_us
 * starting from *elapsed_us (time since the gate last fired), per Timer.hpp:
 * 27-54.  tick_out[i] = 1 if step i fires.  Updates *elapsed_us. */
int afe_plan_ticks(double logic_period_s, uint64_t *elapsed_us, uint64_t dt_us,
                   int n_steps, uint8_t *tick_out);

/* ---- zero-copy device view ---------------------------------------------
 * Raw device pointers to the SoA slabs for HIP-side consumers (renderers,
 * planners, torch via __cuda_array_interface__).  Each field is planar with
 * `stride` elements between components.  state_elem_size is 4 or 8.
 * motor_speed is current as of this call: for stateless motors (tau_m = J_m =
 * 0) driven by held commands the step kernel does not store the rotor speeds
 * (they are clamp(max(0, cmd))); ask for the view again, or use afe_get_state,
 * after further steps.
 * The caller sets struct_bytes = sizeof(afe_device_view) BEFORE the call: the
 * engine fills the members that fit and nothing beyond (a host compiled against
 * an older, shorter struct keeps working); less than the members up to
 * type_index is AFE_ERR_INVALID_ARG.  The call ends a resident step grid first
 * (afe_set_step_mode) and leaves the engine's stream with every authorised
 * step queued before anything the caller orders behind it: slabs read by the
 * caller's own kernels after a stream-ordered wait (or after afe_sync) hold the
 * state after the last afe_step. */
typedef struct afe_device_view {
  size_t struct_bytes;   /* IN: sizeof(afe_device_view) as the caller was compiled */
  int64_t n_vehicles;
  int64_t stride;
  int state_elem_size;
  void *pos, *vel, *att, *ang_vel, *motor_speed; /* 3,3,4,3,4 components */
  void *ext_force, *ext_torque;                  /* 3,3 components */
  float *motor_cmd;                              /* 4 components */
  float *gyro, *acc;                             /* 3,3 components */
  uint32_t *rng;
  uint8_t *type_index;   /* READ ONLY: the engine chooses its step kernel from a host mirror of this
                          * slab (afe_set_vehicle_types keeps both in step); writing it through this
                          * pointer desynchronises them */
  /* pos holds x and y RELATIVE to where they were last set (SetPosition): absolute x = pos_anchor_xy[i] + pos[i],
   * absolute y = pos_anchor_xy[stride + i] + pos[stride + i], added in double; z is absolute.  An fp32 x far from
   * the origin cannot resolve a slow vehicle's motion (at 4 km one ulp is 0.24 mm), an offset from the set point
   * can.  AFE_F64 engines keep absolute positions like the reference: their anchors are zero.  The getters,
   * afe_pack_positions and the depth camera add the anchors themselves. */
  const double *pos_anchor_xy; /* 2 components */
} afe_device_view;
int afe_get_device_view(afe_engine *e, afe_device_view *out);

/* Algorithmic HBM bytes one vehicle-step of the current configuration moves:
 * the slab accesses the step kernel actually makes (SURVEY 8d's 176 B for fp32
 * state + cmds + IMU, minus the rotor-speed read / store the configuration
 * lets it skip, +12 per active wrench array, IMU / RNG / logic bytes on a tick;
 * DESIGN.md section 3); what bench.py's roofline is computed from. */
int afe_algorithmic_bytes_per_step(const afe_engine *e, int imu_tick, double *bytes);

/* Self-test hook for the IMU noise generator: for each of n minstd_rand0 words
 * the first six std::normal_distribution<double>(0,1) draws the engine's
 * device code produces from it (normals6[6*i .. 6*i+5], libstdc++ order) and
 * the engine word afterwards.  Lets a host check the device generator against
 * libstdc++ known answers (tests/golden/rng_kat.json) without going through a
 * physics step.  Replaces nothing in the reference. */
int afe_selftest_normals(afe_engine *e, const uint32_t *seeds, int64_t n,
                         double *normals6, uint32_t *state_after);
/* The same for the AFE_F32 engine's generator: identical engine words and accepted
 * candidates; the polar multiplier and the final product are evaluated in float (the
 * sample is narrowed to float in the reference as well, Quadcopter_T.cpp:167-169),
 * each value within a few float ulp of float(libstdc++'s double). */
int afe_selftest_normals_f32(afe_engine *e, const uint32_t *seeds, int64_t n,
                             float *normals6, uint32_t *state_after);

/* ---- HIP-event timing on the engine's stream (for bench.py) ------------- */
int afe_event_create(void **event);
int afe_event_destroy(void *event);
int afe_event_record(afe_engine *e, void *event);
int afe_event_elapsed_ms(void *start, void *stop, float *ms);

/* ---- multi-GPU shared-world exchange ------------------------------------
 * The step reads no other vehicle (Quadcopter_T.cpp:85-203), so shards step with
 * no data-path collective.  The only inter-vehicle exchange of the path is the
 * shared-world query -- the reference's analogue is UWBNetwork::Run reading other
 * vehicles' true positions (Components/Components/Simulation/UWBNetwork.cpp:
 * 54-84) -- and its payload is the positions, 12 B per vehicle, gathered onto
 * every GPU at query cadence (<= 100 Hz), never per step.
 *
 * afe_pack_positions writes this shard's positions as fp32 planar xyz into a
 * caller-provided DEVICE buffer of 3*n_vehicles floats, on the engine's stream
 * (for hosts that bring their own collective, e.g. torch.distributed). */
int afe_pack_positions(afe_engine *e, float *device_xyz);

/* One process per GPU: afe_comm wraps an RCCL communicator (RCCL is loaded at
 * run time; AFE_ERR_COMM when it is absent).  Rank 0 makes the id, the host
 * hands it to the other ranks by whatever means it has (MPI, a file, a socket,
 * torch.distributed), every rank calls afe_comm_create -- collectively, like
 * ncclCommInitRank. */
typedef struct afe_comm afe_comm;
#define AFE_COMM_ID_BYTES 128
int afe_comm_unique_id(uint8_t id[AFE_COMM_ID_BYTES]);
int afe_comm_create(afe_comm **out, const uint8_t id[AFE_COMM_ID_BYTES], int rank, int n_ranks, int device);
int afe_comm_info(const afe_comm *c, int *rank, int *n_ranks); /* as the communicator reports them */
int afe_comm_destroy(afe_comm *c);
const char *afe_comm_last_error(const afe_comm *c);
/* The all-gather: pack + ncclAllGather (per component, one group) on the engine's
 * stream.  counts[r] = vehicles of rank r, or NULL when every rank holds as many
 * as this one.  dev_xyz_all: DEVICE buffer of 3*n_all floats, planar [3][n_all],
 * vehicles in global order (rank 0's block first).  Asynchronous like afe_step. */
int afe_gather_positions(afe_engine *e, afe_comm *c, const int64_t *counts, float *dev_xyz_all);
/* The exchange behind it with the transport as an argument: which block goes where (counts, offsets; one all-gather per
 * component for equal shards, one broadcast per rank and component for unequal ones), over opaque buffers.  A host that
 * brings its own collective (MPI, torch.distributed) passes it here -- packed_xyz is this rank's [3][n_local] block as
 * afe_pack_positions wrote it, xyz_all the [3][n_all] result, both in whatever memory the transport moves; the
 * callbacks return 0 on success.  afe_gather_positions is this routine with RCCL on device memory; the CPU test suite
 * runs it over gloo on host memory (tests/test_sharding_gloo.py).  Pure host code: usable without a GPU. */
typedef struct afe_gather_transport {
  void *ctx;
  int (*all_gather)(void *ctx, const float *send, float *recv, int64_t count);            /* recv holds n_ranks x count */
  int (*broadcast)(void *ctx, const float *send, float *recv, int64_t count, int root);  /* root's send into everybody's recv */
  int (*group_start)(void *ctx);                                                          /* optional (NULL) */
  int (*group_end)(void *ctx);
} afe_gather_transport;
int afe_gather_exchange(const afe_gather_transport *t, int rank, int n_ranks, const int64_t *counts, int64_t n_local,
                        const float *packed_xyz, float *xyz_all);

/* One process, several GPUs -- the reference's own process model, one loop over a
 * std::vector of vehicles (AIFS_ROS/hiperlab_rostools/src/Simulator/main.cpp:
 * 58-95,323-325): the ensemble is cut into contiguous shards, shard i lives on
 * devices[i] (a device may be listed more than once: logical shards), each with
 * its own engine, stream and RNG words; first_global_index is set so that seeds
 * and neighbour indices are those of the unsharded ensemble.  Results are bit-
 * identical to one engine holding all vehicles. */
typedef struct afe_group afe_group;
int afe_group_create(afe_group **out, int64_t n_vehicles, int precision, const int *devices, int n_devices);
int afe_group_destroy(afe_group *g);
int afe_group_size(const afe_group *g, int *n_shards, int64_t *n_vehicles);
/* shard -> its engine (configure / fill / read it with the afe_* calls above) and global range */
int afe_group_shard(afe_group *g, int shard, afe_engine **engine, int64_t *first, int64_t *count);
int afe_group_step(afe_group *g, uint64_t dt_us, int n_steps); /* afe_step on every shard; launches overlap */
int afe_group_sync(afe_group *g);
/* every shard's positions onto every shard's device by direct peer copies (xGMI):
 * dev_xyz_all_out[i] (optional) = shard i's planar [3][n_vehicles] buffer, owned by the group */
int afe_group_gather_positions(afe_group *g, float **dev_xyz_all_out);
const char *afe_group_last_error(const afe_group *g);
/* *all_pairs = 1 if every pair of the group's devices has direct peer access; 0 if some pair has none
 * (IOMMU, VMs, restricted containers): such a group still works -- its gather then copies row by row with
 * hipMemcpyPeerAsync, which the runtime stages through host memory where it must -- afe_group_create says so once on
 * stderr and the gather is slower.  UNVERIFIED on a machine whose devices really lack peer access (none was available):
 * the staged path itself runs in the test suite, between peers and between logical shards of one device.  (The
 * reference's analogue is one address space for all vehicles: nothing to ask.) */
int afe_group_peer_access(const afe_group *g, int *all_pairs);
/* staged != 0: the gather uses the staged path (above) even between peers -- for a host that knows its direct copies
 * misbehave; 0: back to direct copies (refused with AFE_ERR_INVALID_ARG when a pair has no peer access). */
int afe_group_set_staged_copies(afe_group *g, int staged);

/* ---- shared-world consumers of the gathered buffer -----------------------
 * Nearest neighbour (collision / separation monitoring): for each local vehicle
 * the squared distance (fp32: ((dx*dx + dy*dy) + dz*dz), each operation rounded)
 * to, and global index of, its nearest other vehicle among all_xyz (device, fp32
 * planar, n_all vehicles); the lowest index wins among equally near ones; -1 /
 * 3.4e38 when there is none or the vehicle's own position is not finite.
 * Outputs are DEVICE buffers of n_vehicles entries.  Implementation: uniform
 * grid (counting sort by cell on the device, 3x3x3 search, further rings until
 * no unvisited cell can be closer, isolated vehicles by brute force) -- exact.
 * An engine that is a shard (n_vehicles < n_all) shapes the grid on its own block
 * of all_xyz and sorts only the other shards' vehicles near it; the cost of a query
 * then follows the shard, not the gathered ensemble.  Results do not depend on it.
 * cell_size <= 0 (and afe_nearest_neighbour): chosen from the occupied box,
 * about two vehicles per cell. */
int afe_nearest_neighbour(afe_engine *e, const float *all_xyz, int64_t n_all,
                          float *dist2_out, int32_t *index_out);
int afe_nearest_neighbour_grid(afe_engine *e, const float *all_xyz, int64_t n_all, float cell_size,
                               float *dist2_out, int32_t *index_out);
/* The same query (automatic cell size) without stalling the steps: it runs on a stream of its own, ordered behind what
 * the engine's stream holds at the call (the gather that filled all_xyz), while later afe_step calls proceed.  Whatever
 * would rewrite the gathered buffer or the query's scratch (afe_gather_positions, afe_pack_positions, another query) is
 * ordered behind it on the device; the host joins with afe_query_sync before it reads dist2_out / index_out.
 * (Measured, DESIGN.md section 5: the query and the steps both live on the memory system, so running them together
 * hides the query's launch gaps, not its traffic.) */
int afe_nearest_neighbour_async(afe_engine *e, const float *all_xyz, int64_t n_all, float *dist2_out, int32_t *index_out);
int afe_query_sync(afe_engine *e);
/* How often the grid is re-shaped from the ensemble's current bounds and spread: every
 * `every_n_queries` queries (default 1).  Re-shaping needs one 6 KB read-back, i.e. it synchronises
 * the stream; in between, queries are fully asynchronous.  Results do not depend on it (the query is
 * exact for any grid; vehicles that have left the grid's core are clamped into its boundary cells). */
int afe_set_neighbour_grid_refresh(afe_engine *e, int every_n_queries);
/* How often the vehicles are sorted into the grid's cells: every `every_n_queries` queries (default 1).  In between a
 * query keeps the cell ORDER of the last sort and only refreshes the positions in it (one pass instead of the five of a
 * counting sort); the device keeps a bound on how far any vehicle has moved since the sort and the search stops a ring
 * only where that bound allows -- the answers are exactly the sorted query's whatever the vehicles did (vehicles moving
 * centimetres between queries through cells metres wide cost nothing; an ensemble that has moved by cells searches wider
 * rings, and one that has jumped is answered by the brute force until the next sort).  A grid re-shape sorts anyway. */
int afe_set_neighbour_sort_reuse(afe_engine *e, int every_n_queries);
/* shape of the grid the last query used, and how many of its queries fell through to brute force
 * (isolated vehicles; reading that count synchronises the device).  Any pointer may be NULL. */
int afe_neighbour_grid_info(const afe_engine *e, int dims[3], float *cell_size, int64_t *n_cells, int64_t *n_bruteforce);
/* The O(n_queries * n_all) definition itself, for listed local vehicles (DEVICE
 * array of local indices): the cross-check of the grid at full ensemble size.
 * dist2_out / index_out are indexed by local vehicle like above. */
int afe_nearest_neighbour_bruteforce(afe_engine *e, const float *all_xyz, int64_t n_all, const int32_t *dev_queries,
                                     int64_t n_queries, float *dist2_out, int32_t *index_out);

/* Measurement aid (bench.py's `roofline.peak_measured`; SURVEY 8d asks for the box's measured streaming figure next
 * to the nominal peak): `launches` back-to-back launches of a kernel in the step kernel's own launch shape -- one-wave
 * workgroups, one lane per element, n_read planar dword read streams and n_write write streams in place through one
 * buffer resource, slab stride 256 x odd, no arithmetic to speak of -- on a scratch buffer of its own; returns the
 * best of three event-timed runs.  (n_read, n_write) = (20, 13): the 132 bytes of the off-tick step launch,
 * (24, 17): the 164 bytes of the tick launch.  No reference counterpart. */
int afe_stream_probe(int device, int64_t n, int n_read, int n_write, int launches, float *us_per_launch);

/* UWB ranging network: Simulation::UWBNetwork (Components/Components/Simulation/
 * UWBNetwork.hpp:16-50, UWBNetwork.cpp:8-89).  afe_uwb_create = the constructor's
 * rng.seed(0) (:19); afe_uwb_set_noise = SetNoiseProperties (UWBNetwork.hpp:28-33).
 * afe_uwb_range completes n_pairs ranging transactions in order, each exactly as
 * Run() does (:66-71): one uniform draw decides outlier or not, one normal draw
 * gives the noise, range = float(|p_requester - p_responder| + noise) or
 * float(normal * outlierStdDev) -- same generator and distribution classes
 * (std::mt19937, uniform_real_distribution, normal_distribution incl. its cached
 * second value), one stream for the network's lifetime (the reference's are file-scope objects shared by every
 * UWBNetwork of the process, UWBNetwork.cpp:4-6: the first network of a process matches it draw for draw; a process
 * with several networks is reproduced by drawing for all of them from ONE afe_uwb_network).  The true positions come
 * from the gathered buffer (requester / responder are GLOBAL vehicle indices, host
 * arrays); the norm is evaluated in double like Vec3d::GetNorm2.  range_out
 * (host, n_pairs floats) is what every radio "hears" (:77-80); outlier_out is
 * optional.  afe_uwb_draw exposes the stream alone (host only, no GPU). */
typedef struct afe_uwb_network afe_uwb_network;
int afe_uwb_create(afe_uwb_network **out);
void afe_uwb_destroy(afe_uwb_network *u);
int afe_uwb_set_noise(afe_uwb_network *u, double noise_std_dev, double outlier_probability, double outlier_std_dev);
int afe_uwb_draw(afe_uwb_network *u, int64_t n_pairs, double *noise_term, uint8_t *is_outlier);
int afe_uwb_range(afe_uwb_network *u, afe_engine *e, const float *dev_all_xyz, int64_t n_all,
                  const int32_t *requester, const int32_t *responder, int64_t n_pairs, float *range_out,
                  uint8_t *outlier_out);

#ifdef __cplusplus
}
#endif
#endif /* AGRIFLY_ENGINE_H */
