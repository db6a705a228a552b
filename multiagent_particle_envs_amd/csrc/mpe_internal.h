// mpe_internal.h -- launch entry points shared between the kernel translation units and the C ABI.
#pragma once
#include "mpe_device.h"

namespace mpe {

enum class NarrowOp { Step, Observe };

// thread-per-world family (mpe_narrow.hip)
bool narrow_supports(int kind, int A, int L, int nadv);
int launch_narrow(NarrowOp op, int kind, int A, int L, int nadv, const NarrowDesc &d, const MpeBuffers &b,
                  size_t B, hipStream_t stream);
int launch_phase(int phase, int A, int L, const NarrowDesc &d, const MpeBuffers &b, size_t B,
                 hipStream_t stream);

// wave-per-world family for large entity counts (mpe_wide.hip)
struct WideDesc {
  int32_t kind, A, L, dim_c, collaborative;
  int32_t nadv;  // simple_tag: agents [0, nadv) are adversaries
  int32_t D;  // obs width (spread: same for every agent)
  float dt, damp, cforce, cmargin, cmargin_inv;
  // homo: every agent has the same constants (below) and no landmark collides -- the shipped simple_spread at any N;
  // kernels specialised for it take the constants from here instead of the device table
  int32_t homo, a_flags /* 1 movable | 2 collide */;
  float a_size, a_inv_mass, a_accel, a_max_speed;
  int32_t rows_nt;  // set by launch_wide: every wave store of rows is a whole number of 128-byte lines -> nontemporal stores
};
constexpr int kEntityTableCols = 6;  // size, mass, accel, max_speed, movable, collide  -> [6][E] floats
struct RollArgs;
bool wide_supports(const WideDesc &d, bool out);
int launch_wide(bool phys, bool out, const WideDesc &d, const MpeBuffers &b, size_t B, hipStream_t stream,
                const RollArgs *roll = nullptr);

// reset / synthetic actions / fused rollout (mpe_rng.hip)
int launch_reset(int A, int L, const MpeBuffers &b, size_t B, const uint8_t *mask, float landmark_range,
                 uint64_t seed, uint64_t episode, uint64_t world_offset, int n_choices, const int32_t *pop,
                 hipStream_t stream);
struct ResetBoxes { float box[MPE_ROWS_MAX_ENTITIES][4]; };
int launch_reset_box(int A, int L, const MpeBuffers &b, size_t B, const uint8_t *mask, const ResetBoxes &boxes, uint64_t seed,
                     uint64_t episode, uint64_t world_offset, int n_choices, const int32_t *pop, hipStream_t stream);
int launch_episode_tick(int32_t *episode_step, uint8_t *done, int A, size_t B, int max_steps, int clear_finished,
                        hipStream_t stream);
int launch_reset_random_actions(int A, int L, const MpeBuffers &b, size_t B, float landmark_range, uint64_t episode, int n_choices,
                                const int32_t *pop, float *act, int32_t *ids, uint64_t seed, uint64_t step0, int T,
                                uint64_t world_offset, hipStream_t stream);
int launch_random_actions(float *act, int32_t *ids, int A, size_t B, uint64_t seed, uint64_t step0, int T,
                          uint64_t world_offset, hipStream_t stream);

int launch_random_comm(float *comm, int A, size_t B, int dim_c, unsigned speakers, uint64_t seed, uint64_t step0, int T,
                       uint64_t world_offset, hipStream_t stream);

// wave-per-agent / lane-per-world family (mpe_split.hip): fused step and fused T-step rollout
struct RollArgs {
  int32_t T;            // steps in this launch (ignored by the single-step kernel)
  int32_t episode_len;  // in-kernel reset every episode_len global steps; 0 = never
  int32_t trajectory;   // outputs of step t go to block t of the output buffers (else: overwrite block 0)
  int32_t observe_only; // single-step kernel: skip World.step, emit the outputs of the current state (mpe_observe)
  int32_t wpw;          // worlds per workgroup of the wave-per-agent kernels (set by launch_split)
  float landmark_range;
  uint64_t seed, step0, world_offset;
  const float *act_seq;   // rollouts of the wide / row-program kernels: the CALLER's moves, T consecutive [A][B][5] tensors (step t of
                          // the launch reads tensor t) instead of moves drawn in the kernel; nullptr = drawn (mpe_rollout_random / _rows)
};
bool split_supports(int kind, int A, int L, int nadv);
// the step server (mpe_split.hip, SERVE): device words and ring geometry of one server (mpe_step_server_*)
struct ServeHandles {
  uint64_t *door, *flag;
  uint32_t *status;
  const float *act_ring, *comm_ring;
  int32_t ring, slots;
  uint64_t timeout_ticks;
  bool ahead;              // every command precedes its launch: no residency needed
};
constexpr int MPE_ESERVER_TOO_LARGE = -1000;   // internal: the grid cannot be resident (mapped to MPE_EUNSUPPORTED with a message)
bool serve_supports(int kind, int A, int L, int nadv);
unsigned serve_grid(size_t B);
int launch_serve_ring(uint64_t *door, uint64_t n, hipStream_t stream);
int launch_serve_wait(const uint64_t *flag, unsigned n_flags, uint64_t completed, uint32_t *status, uint64_t timeout_ticks,
                      hipStream_t stream);
int launch_split_serve(int kind, int A, int L, int nadv, const NarrowDesc &d, const MpeBuffers &b, size_t B, const RollArgs &ra,
                       const ServeHandles &h, hipStream_t stream);
int launch_split(bool roll, int kind, int A, int L, int nadv, const NarrowDesc &d, const MpeBuffers &b, size_t B,
                 const RollArgs &ra, hipStream_t stream);

// the composable output stage (mpe_rows.hip): kernel-side header of an MpeRowProgram
constexpr int kRowSlots = 8;
constexpr int kRowPicks = MPE_MAX_CHOICES;
constexpr int kRowMaxObsWaves = 16;   // 1024 threads
constexpr int kRowSelf = MPE_ROW_SELF;
enum {
  ROW_OBS_VEL = MPE_ROW_OBS_VEL, ROW_OBS_POS = MPE_ROW_OBS_POS, ROW_OBS_REL = MPE_ROW_OBS_REL, ROW_OBS_REL_PICK = MPE_ROW_OBS_REL_PICK,
  ROW_OBS_COMM = MPE_ROW_OBS_COMM, ROW_OBS_CONST = MPE_ROW_OBS_CONST, ROW_OBS_ONEHOT = MPE_ROW_OBS_ONEHOT,
  ROW_OBS_REL_VIS = MPE_ROW_OBS_REL_VIS, ROW_OBS_VEL_VIS = MPE_ROW_OBS_VEL_VIS, ROW_OBS_IN_REGION = MPE_ROW_OBS_IN_REGION,
  ROW_OBS_REL_RANGE = MPE_ROW_OBS_REL_RANGE, ROW_OBS_VEL_RANGE = MPE_ROW_OBS_VEL_RANGE, ROW_OBS_REL_VIS_RANGE = MPE_ROW_OBS_REL_VIS_RANGE,
  ROW_OBS_VEL_VIS_RANGE = MPE_ROW_OBS_VEL_VIS_RANGE, ROW_OBS_CONST_N = MPE_ROW_OBS_CONST_N,
  ROW_R_MIN_D2_RANGE = MPE_ROW_R_MIN_D2_RANGE, ROW_R_MIN_D2_TO_RANGE = MPE_ROW_R_MIN_D2_TO_RANGE, ROW_R_ADD_IF_HIT_GRID = MPE_ROW_R_ADD_IF_HIT_GRID, ROW_R_ADD_MIN_DIST_GRID = MPE_ROW_R_ADD_MIN_DIST_GRID,
  ROW_R_D2 = MPE_ROW_R_D2, ROW_R_MIN_D2 = MPE_ROW_R_MIN_D2, ROW_R_D2_PICK = MPE_ROW_R_D2_PICK, ROW_R_MIN_D2_PICK = MPE_ROW_R_MIN_D2_PICK,
  ROW_R_SQRT = MPE_ROW_R_SQRT, ROW_R_BOUND = MPE_ROW_R_BOUND, ROW_R_COMM_ERR = MPE_ROW_R_COMM_ERR, ROW_R_COMM_SUM = MPE_ROW_R_COMM_SUM,
  ROW_R_CONST = MPE_ROW_R_CONST, ROW_R_SAVE = MPE_ROW_R_SAVE, ROW_R_LOAD = MPE_ROW_R_LOAD, ROW_R_ZERO = MPE_ROW_R_ZERO,
  ROW_R_ADD = MPE_ROW_R_ADD, ROW_R_ADD_IF_HIT = MPE_ROW_R_ADD_IF_HIT, ROW_R_ADD_ACC = MPE_ROW_R_ADD_ACC, ROW_R_STORE = MPE_ROW_R_STORE,
  ROW_R_ABS_POS = MPE_ROW_R_ABS_POS, ROW_R_DONE_IF_GT = MPE_ROW_R_DONE_IF_GT, ROW_R_DONE_IF_LT = MPE_ROW_R_DONE_IF_LT, ROW_R_DONE_IF_HIT = MPE_ROW_R_DONE_IF_HIT,
  ROW_OBS_CODE = MPE_ROW_OBS_CODE, ROW_R_CODE = MPE_ROW_R_CODE, ROW_R_DONE_CODE = MPE_ROW_R_DONE_CODE
};
// Scalars of a program: kernel arguments (one batch of scalar loads at wave start).
struct RowDims {
  int32_t n_agents, n_entities, n_vel, dim_c, collaborative;
  int32_t d_max;                 // widest observation row (floats): the waves' tile size
  int32_t n_picks;               // rows of MpeBuffers.choice (desc->n_choices)
  int32_t n_ops;                 // 16-byte ops
  int32_t n_regions, region_entity[2];
  uint32_t all_seeing;
  uint64_t movable, collide;     // bit e
  float dt, damp, cforce, cmargin, cmargin_inv;
  int32_t reset_boxes;           // 1: restarts place entity e in RowTables.reset_box[e] (else: agents [-1,1)^2, landmarks [-r,r)^2)
  int32_t n_shared;              // traced programs: values several agents' rewards share, computed once per world (traced_shared)
};
// Tables of a program: DEVICE memory (uploaded by launch_rows_header whenever their content changes), read by scalar loads;
// a compiled program (MPE_ROWS_STATIC) carries them as constants.
struct RowTables {
  float size[MPE_ROWS_MAX_ENTITIES], inv_mass[MPE_ROWS_MAX_ENTITIES], accel[MPE_ROWS_MAX_ENTITIES], max_speed[MPE_ROWS_MAX_ENTITIES];
  int32_t obs_off[MPE_ROWS_MAX_ENTITIES + 1];     // prefix sums of the row widths
  int32_t obs_begin[MPE_ROWS_MAX_ENTITIES + 1];   // agent i's observation ops: [obs_begin[i], obs_begin[i + 1])
  int32_t rew_begin[MPE_ROWS_MAX_ENTITIES + 1];   // agent i's reward ops
  int32_t done_begin[MPE_ROWS_MAX_ENTITIES + 1];  // agent i's done ops (all equal: none)
  float reset_box[MPE_ROWS_MAX_ENTITIES][4];      // lo_x, span_x, lo_y, span_y of entity e's reset placement (RowDims.reset_boxes)
};
// episode bookkeeping + masked reset in front of the rows (mpe_episode_finish)
struct RowEpisode {
  int32_t enabled;               // 0 off; 1 mpe_episode_finish (decide at entry from the done rows, rows only); 2 mpe_step_rows_episode
                                 // (decide after the step's own done programs, restart inside the launch)
  int32_t max_steps;
  int32_t *episode_step;
  float landmark_range;
  int32_t n_choices, choice_pop[MPE_MAX_CHOICES];
  uint64_t seed, episode, world_offset;
  uint32_t speakers;             // (mpe_rollout_rows) bit a: agent a says a drawn word every step
  uint32_t pad_;
};
int launch_rows_header(const RowTables &t, void *dst, hipStream_t stream);
int launch_rows(const MpeBuffers &b, const RowDims &dims, const RowTables &host, const void *tables_device, bool phys, int vec4,
                const RowEpisode &ep, const int32_t *ops_device, size_t B, hipStream_t stream, const RollArgs *roll = nullptr);
int rows_geometry(const RowDims &dims, bool phys, int *waves, size_t *lds_bytes, int max_waves);
int launch_rows_image(void *const fns[5], const MpeBuffers &b, const RowDims &dims, const RowTables &host, bool phys, int vec4,
                      const RowEpisode &ep, size_t B, hipStream_t stream, const RollArgs *roll = nullptr);

}  // namespace mpe
