/*
 * riab_hip.h — C ABI of libriab_hip.so: the MI355X (gfx950) kernels behind the
 * batched RatInABox hot path.
 *
 * The reference (RatInABox, pure Python/NumPy) has no FFI; this ABI is the
 * drop-in boundary a maintainer would bind with ctypes.
 * Each entry point names the reference code it replaces (paths relative to the
 * reference checkout).
 *
 * Conventions
 *  - every `const T*` / `T*` marked "device" is a device (HBM) pointer owned by
 *    the caller (e.g. a PyTorch-ROCm tensor's data_ptr()); structs are host
 *    memory, read before the call returns.
 *  - nothing is allocated, copied host<->device or synchronised inside; kernels
 *    are enqueued on `stream` (a hipStream_t) and the call returns immediately,
 *    so every entry point is hipGraph-capturable.
 *  - batch axis (agents / positions) is always the fastest-varying axis.
 *    B must be a multiple of 4 and row pointers 16-byte aligned (the host layer
 *    pads the agent axis); rows are `ld` elements apart where stated.
 *  - return: 0 ok; negative = argument error detected before launch
 *    (RIAB_E*); positive = hipError_t of the launch.
 *  - re-entrant: no global state.
 */
#ifndef RIAB_HIP_H
#define RIAB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RIAB_ABI_VERSION 8
#define RIAB_MAX_WALLS 64     /* walls staged in LDS by the motion / BVC / line-of-sight kernels */
#define RIAB_MAX_TEST_ANGLES 360
#define RIAB_STATE_ROWS 12    /* rows of the agent state matrix, see below */
#define RIAB_HIST_ROWS 8      /* rows of one trajectory-history record */
#define RIAB_MAX_BOUNCES 16   /* bound on the reference's `while True` bounce loop (Agent.py:426) */

enum {
  RIAB_OK = 0,
  RIAB_EINVAL = -1,       /* null pointer / negative size / bad enum */
  RIAB_EALIGN = -2,       /* B % 4 != 0 or misaligned row pointer */
  RIAB_ETOOBIG = -3,      /* more walls / test angles than the LDS staging allows */
  RIAB_EUNSUPPORTED = -4, /* combination not implemented on device */
  RIAB_EFULL = -5,        /* a step plan's history chunk has no free row left */
  RIAB_EPARTIAL = -6,     /* riab_simulate_*: a launch failed AFTER the trajectory kernel had been launched: the agent
                             state and the trajectory rows have advanced T steps, the rates of this call are incomplete */
  RIAB_ECHANGED = -7      /* riab_simulate: a watched host array (RiabSimulate.watch) no longer equals its snapshot: the
                             caller's cached device tables are stale; nothing was launched */
};

typedef void* riab_stream_t; /* hipStream_t */

/* Rows of the agent state matrix `state[RIAB_STATE_ROWS][B]` (float64, device).
 * Mirrors the attributes Agent.update mutates (Agent.py:193-242). */
enum {
  RIAB_S_POS_X = 0, RIAB_S_POS_Y = 1,
  RIAB_S_VEL_X = 2, RIAB_S_VEL_Y = 3,          /* Agent.velocity */
  RIAB_S_ROT_VEL = 4,                          /* Agent.rotational_velocity */
  RIAB_S_MVEL_X = 5, RIAB_S_MVEL_Y = 6,        /* Agent.measured_velocity */
  RIAB_S_MROT_VEL = 7,                         /* Agent.measured_rotational_velocity */
  RIAB_S_HD_X = 8, RIAB_S_HD_Y = 9,            /* Agent.head_direction */
  RIAB_S_DIST = 10,                            /* Agent.distance_travelled */
  RIAB_S_DWALL = 11                            /* Agent.distance_to_closest_wall */
};

/* Rows of one history record `hist[t][RIAB_HIST_ROWS][B]` (float32, device):
 * what Agent.save_to_history appends per step (Agent.py:509-521). */
enum {
  RIAB_H_POS_X = 0, RIAB_H_POS_Y = 1,
  RIAB_H_VEL_X = 2, RIAB_H_VEL_Y = 3,          /* history["vel"] = measured velocity */
  RIAB_H_HD_X = 4, RIAB_H_HD_Y = 5,
  RIAB_H_ROT_VEL = 6,                          /* history["rot_vel"] = measured rotational velocity */
  RIAB_H_DIST = 7                              /* history["distance_travelled"] */
};

/* Environment geometry read by the hot path (Environment.py:65-191, 657-894): a 2D box (solid or
 * periodic) or a simple polygon (solid), optionally with polygonal holes.  `walls` is Environment.walls
 * flattened to [n_walls][4] = (ax, ay, bx, by), reference order (Environment.py:128-163): the
 * boundary's edges first (when solid), then the user's walls, then the holes' edges.
 * "Inside the environment" (Environment.check_if_position_is_in_environment, :781-818) is the STRICT
 * interior of the boundary polygon minus the strict interiors of the holes: even-odd crossings over
 * walls[0 : n_boundary] and over the walls flagged in hole_mask (holes are disjoint, so one crossing count
 * serves them all; hole edges may sit anywhere behind the boundary's: Environment.add_wall / add_hole
 * append in call order, Environment.py:330-364); a point on a boundary edge is outside, a point on a hole's
 * edge inside. */
typedef struct RiabEnv {
  double extent[4];      /* left, right, bottom, top (the bounding box of the boundary) */
  double scale;          /* Environment.scale (periodic wrap length, Environment.py:670-674) */
  int32_t periodic;      /* boundary_conditions == "periodic" */
  int32_t n_walls;
  const double* walls;   /* device, float64 [n_walls][4] */
  int32_t polygon;       /* the boundary is not the rectangle `extent` (Environment.is_rectangular == False) */
  int32_t n_boundary;    /* boundary edges at the head of `walls` (4 for a solid box; only read when polygon) */
  uint64_t hole_mask;    /* bit k: walls[k] is an edge of a hole (RIAB_MAX_WALLS = 64 walls; 0: no holes) */
} RiabEnv;

/* Motion parameters of one Agent.update call, resolved on the host:
 * `*_kw` are the values after per-call kwargs overrides (Agent.py:280-285,
 * 353-355), plain names are the Agent attributes the reference reads
 * regardless of kwargs (Agent.py:310, 340, 375, 439, 489). */
typedef struct RiabMotion {
  double dt;
  double rot_theta_kw;        /* 1 / rotational_velocity_coherence_time */
  double rot_sigma_kw;        /* sqrt(2 std^2 / (tau dt)), utils.py:365 */
  double rot_drift_kw;        /* kwarg rotational_velocity_drift, default 0 */
  double speed_theta_kw;      /* 1 / speed_coherence_time */
  double speed_sigma_kw;      /* sqrt(2 / (tau dt)) */
  double speed_mean_kw;       /* Rayleigh sigma in the stochastic update */
  double speed_mean;          /* attribute: wall spring speed, bounce speed */
  int32_t speed_std_is_zero;  /* attribute speed_std == 0 (Agent.py:310) */
  int32_t has_drift;          /* drift_velocity given */
  double drift_theta;         /* ratio / speed_coherence_time (attribute), Agent.py:340 */
  double wall_repel_strength_kw;
  double wall_repel_distance_kw;
  double thigmotaxis_kw;
  double hd_tau;              /* head_direction_smoothing_timescale (attribute) */
  /* Optional broad phase for rooms with many walls (NULL: every wall is looked at every step).  A grid of
   * wall_grid_n x wall_grid_n cells over the extent; per cell two 64-bit masks of walls: [0] every wall that can be the
   * NEAREST wall of, or within wall_grid_wd of, a point of the cell; [1] every wall within wall_grid_lmax of the cell
   * (the walls a step of at most that length from a point of the cell can cross).  Conservative supersets, built by
   * the caller (ratinabox_amd/Environment.py: wall_grid): the motion step evaluates its per-wall arithmetic for the
   * walls of the mask only — the minimum, the near set and the first-hit wall are the same, bit for bit.  Ignored when
   * wall_repel_distance_kw > wall_grid_wd; a step longer than wall_grid_lmax looks at every wall. */
  const uint64_t* wall_grid;  /* device uint64 [wall_grid_n * wall_grid_n][2], cell (ix, iy) at iy * wall_grid_n + ix */
  int32_t wall_grid_n;        /* <= RIAB_WALL_GRID_MAX */
  double wall_grid_wd;
  double wall_grid_lmax;
} RiabMotion;
#define RIAB_WALL_GRID_MAX 16

/* T fused Agent.update() steps for B independent agents.
 * Replaces Agent.update (Agent.py:160-242) = _stochastic_velocity_update
 * (:268-322) + _drift_velocity_update (:324-341) + _wall_velocity_update
 * (:343-421) + _check_and_handle_wall_collisions (:423-441) + boundary safety
 * net (:221-222, Environment.py:781-894) + _measure_velocity_of_step_taken
 * (:444-472) + _update_head_direction (:474-500) + _update_distance_travelled
 * (:502-507) + save_to_history (:509-521), with utils.py:30-184, 231-368,
 * 409-421 inlined.
 *
 *  state      device float64 [RIAB_STATE_ROWS][B], read and written
 *  agent_id0  global id of agent 0 of this shard (keys the RNG; results do not
 *             depend on how agents are sharded over GPUs)
 *  drift      device float64 [2][B] or NULL (drift_velocity per agent)
 *  z_in       device float64 [T][2][B] or NULL: the two standard normals the
 *             reference draws per update (rotation OU, speed OU).  NULL =>
 *             Philox4x32-10 keyed by (seed; step0+t, agent id)
 *  z_out      device float64 [T][2][B] or NULL: records the normals used
 *  forced_pos device float64 [T][2][B] or NULL: imported / forced positions — the motion
 *             model is skipped, the step is "moved to forced_pos[t]" and velocity /
 *             rotational velocity are overwritten by the measured ones
 *             (Agent._update_position_along_imported_trajectory / forced_next_position,
 *             Agent.py:229-238, 244-266)
 *  resample_pos device float64 [T][2][B] or NULL: where an agent that ends a step outside a polygonal
 *             boundary or inside a hole is put (Environment.apply_boundary_conditions' resample branch,
 *             Environment.py:886-893: the reference draws `sample_positions(n=1, method="random")` from
 *             np.random; parity runs hand the accepted positions in).  NULL => uniform draws over
 *             `extent` from Philox(seed; step, agent id, attempt), rejected until inside (at most 64)
 *  hist       device float32 [T][RIAB_HIST_ROWS][B] or NULL
 *  diag       device int32 [4] or NULL, atomically accumulated:
 *             [0] bounces, [1] saturations of the bounded loops (bounces, resample attempts), [2] boundary
 *             conditions applied, [3] zero-displacement steps
 * Arithmetic is float64, like the reference's: on gfx950 a float64 FMA issues at the float32 rate, and the float32
 * variant this entry point had in rounds 1-2 (library exp / log / erf in float32) was SLOWER than the table-driven
 * float64 path (2.9 vs 1.8 us per step) and 1e-5 instead of 1e-12 accurate: removed in ABI v4.
 */
int riab_agent_step(const RiabEnv* env, const RiabMotion* motion, double* state, int64_t B,
                    int64_t agent_id0, const double* drift, const double* z_in, double* z_out,
                    const double* forced_pos, const double* resample_pos, uint64_t seed, uint64_t step0, int32_t T,
                    float* hist, int32_t* diag, riab_stream_t stream);

/* ---- Environment geometry queries ---------------------------------------------------------
 * The stand-alone forms of the helpers riab_agent_step / riab_place_cells inline, for callers of
 * the reference's Environment API.  float64 like the reference's functions; device pointers;
 * `geometry` is RIAB_GEOM_* (declared below). */

/* Environment.get_vectors_between___accounting_for_environment (Environment.py:657-675) and
 * get_distances_between___accounting_for_environment (Environment.py:677-779) for all pairs
 * (pos1[i], pos2[j]): vec = pos1[i] - pos2[j], wrapped when periodic; dist = |vec|, 1000 where
 * an internal wall (walls[4:]) blocks the line of sight (line_of_sight), or the route round the
 * single internal wall (geodesic).  x1,y1 [N1]; x2,y2 [N2]; dist / vec_x / vec_y [N1][N2], each
 * may be NULL (not all). */
int riab_env_pairwise(const RiabEnv* env, const double* x1, const double* y1, int64_t N1, const double* x2,
                      const double* y2, int64_t N2, int32_t geometry, double* dist, double* vec_x, double* vec_y,
                      riab_stream_t stream);

/* Environment.vectors_from_walls (Environment.py:843-853) = utils.shortest_vectors_from_points_to_lines
 * (utils.py:121-184) for P positions: out [n_walls][2][P], the vector from the nearest point of
 * wall w to position p. */
int riab_env_vectors_from_walls(const RiabEnv* env, const double* pos_x, const double* pos_y, int64_t P,
                                double* out, riab_stream_t stream);

/* Environment.check_wall_collisions (Environment.py:820-841; utils.vector_intercepts with
 * return_collisions=True, utils.py:74-106) for P proposed steps (x0,y0)->(x1,y1):
 * out uint8 [n_walls][P], 1 where the step strictly crosses the wall. */
int riab_env_check_wall_collisions(const RiabEnv* env, const double* x0, const double* y0, const double* x1,
                                   const double* y1, int64_t P, uint8_t* out, riab_stream_t stream);

/* Environment.check_if_position_is_in_environment (Environment.py:781-818: strict interior of the
 * boundary, outside every hole) -> inside_out uint8 [P] (or NULL), and, when `apply`,
 * Environment.apply_boundary_conditions (Environment.py:855-894) in place for a rectangular environment:
 * positions outside the box are clamped to [min+0.01, max-0.01] (solid) or wrapped modulo the extent
 * (periodic).  Positions that need the RESAMPLE branch (inside a hole, or outside a polygonal boundary) are
 * left unchanged and flagged 2 in inside_out: the caller draws their replacements (the reference draws them
 * from np.random). */
int riab_env_boundary_conditions(const RiabEnv* env, double* pos_x, double* pos_y, int64_t P, uint8_t* inside_out,
                                 int32_t apply, riab_stream_t stream);

/* Where a firing-rate kernel reads positions and writes rates / spikes.
 * Positions are T rows of B agents: row t of x starts at pos_x + t*pos_ld
 * (so a trajectory history [T][8][B] is consumed in place with pos_ld = 8*B,
 * and a plain list of P positions is T=1, B=P).  Output is rates[t][c][b]
 * = Neurons.history["firingrate"][t][c] of agent b; for T=1 it is the
 * reference's get_state() layout (n_cells, P) (Neurons.py:943-949). */
typedef struct RiabRateIO {
  const float* pos_x;    /* device */
  const float* pos_y;    /* device */
  const float* hd_x;     /* device or NULL: head direction rows (HDC, egocentric BVC) */
  const float* hd_y;
  int64_t pos_ld;        /* elements between consecutive time rows */
  int64_t T, B;
  float* rates;          /* device float32 [T][n][B] */
  uint8_t* spikes;       /* device uint8 [T][n][B] or NULL (Neurons.py:681-687) */
  const float* u_in;     /* device float32 [T][n][B] or NULL: explicit spike uniforms */
  float dt;              /* Agent.dt for the spike rule `u < dt*rate` */
  float min_fr, max_fr;  /* rates are scaled to [min_fr, max_fr] */
  uint64_t seed;         /* Philox key for spikes when u_in == NULL */
  uint64_t step0;        /* global index of time row 0 */
  int64_t agent_id0;     /* global id of agent 0 (multiple of 4) */
  int32_t pop_id;        /* distinguishes Neurons populations in the RNG stream */
} RiabRateIO;

enum { RIAB_PC_GAUSSIAN = 0, RIAB_PC_GAUSSIAN_THRESHOLD = 1, RIAB_PC_DIFF_OF_GAUSSIANS = 2,
       RIAB_PC_ONE_HOT = 3, RIAB_PC_TOP_HAT = 4 };
enum { RIAB_GEOM_EUCLIDEAN = 0, RIAB_GEOM_LINE_OF_SIGHT = 1, RIAB_GEOM_GEODESIC = 2 };

/* PlaceCells.get_state (Neurons.py:936-981) with
 * Environment.get_distances_between___accounting_for_environment
 * (Environment.py:677-779).  cells device float32 [n][3] = (centre x, centre y,
 * k = -log2(e)/(2 w^2)) per cell (host, float64, rounded once) so that the
 * gaussian is exp2(d^2 * k).  top_hat_width is the scalar `widths`
 * parameter the reference's top_hat compares against (Neurons.py:976).
 * line_of_sight / geodesic read env->walls[4:] (Environment.py:715-717). */
int riab_place_cells(const RiabEnv* env, const RiabRateIO* io, const float* cells, int32_t n,
                     int32_t description, int32_t geometry, float top_hat_width,
                     riab_stream_t stream);

/* RandomSpatialNeurons.get_state (Neurons.py:2916-2960): the kernel-weighted local average of the
 * sampled targets, rate[c][p] = sum_m k(p, X_m) targets[m][c] / sum_m k(p, X_m) with
 * k = exp(-d^2 / (2 lengthscale^2)) and d the environment distance of riab_place_cells.
 *  anchors device float32 [M][3] = (X_m x, X_m y, -log2(e) / (2 lengthscale^2))
 *  targets device float32 [M][n], already squashed into [min_fr, max_fr] (Neurons.py:2911-2912);
 *          io->min_fr / max_fr are not applied again. */
int riab_random_spatial_neurons(const RiabEnv* env, const RiabRateIO* io, const float* anchors, int32_t M,
                                const float* targets, int32_t n, int32_t geometry, riab_stream_t stream);

enum { RIAB_GC_RECTIFIED = 0, RIAB_GC_SHIFTED = 1 };

/* GridCells.get_state, 2D (Neurons.py:1172-1236).  table device float32 [n][9]:
 * entries 3i..3i+2 of a cell = (a_i, bx_i, by_i) with the phase of cosine i, in
 * revolutions, phi_i/2pi = a_i - (x*bx_i + y*by_i); built on the host in
 * float64 from gridscales, phase_offsets and w (Neurons.py:1154-1161, 1192-1203).
 * f0 = firing_rate_at_full_width (Neurons.py:1211). */
int riab_grid_cells(const RiabRateIO* io, const float* table, int32_t n, int32_t description,
                    float f0, riab_stream_t stream);

/* HeadDirectionCells.get_state, 2D (Neurons.py:2421-2485): von Mises of
 * utils.get_angle(head_direction).  table device float32 [n][3] = (cosine and sine of the
 * preferred angle, log2(e)/sigma^2) per cell.  Needs io->hd_x / hd_y. */
int riab_head_direction_cells(const RiabRateIO* io, const float* table, int32_t n,
                              riab_stream_t stream);

/* VelocityCells.get_state (Neurons.py:2577-2583): the HeadDirectionCells tuning of the NORMALISED
 * velocity, scaled to [min_fr, max_fr] and then multiplied by |v| / one_sigma_speed
 * (one_sigma_speed = Agent.speed_mean + Agent.speed_std at construction, Neurons.py:2567).
 * The velocity is read from vel_x / vel_y: device float64 [B], rows RIAB_S_VEL_X / _Y of the agent
 * state (what `evaluate_at="agent"` reads: Agent.velocity, not the measured velocity; T must be 1),
 * or, when they are NULL, from the float32 rows io->hd_x / hd_y (any T). */
int riab_velocity_cells(const RiabRateIO* io, const float* table, int32_t n, float one_sigma_speed,
                        const double* vel_x, const double* vel_y, riab_stream_t stream);

/* SpeedCell.get_state (Neurons.py:2632-2651): one cell, |v| / one_sigma_speed scaled to
 * [min_fr, max_fr]; v = the float32 rows io->hd_x / hd_y (the newest history["vel"] = measured
 * velocity at the agent: rows RIAB_H_VEL_X / _Y).  rates is [T][1][B]. */
int riab_speed_cell(const RiabRateIO* io, float one_sigma_speed, riab_stream_t stream);

/* BoundaryVectorCells.get_state (Neurons.py:1617-1778) with utils.vector_intercepts
 * (utils.py:30-118), gaussian / von_mises (utils.py:424-457).
 *  test_dirs  device float64 [K][2] unit test directions (Neurons.py:1584-1596)
 *  ray_rden   device float64 [K][n_walls]: 1 / (u_k . s_w^perp) = 1 / (ux*(-sy) + uy*sx), the
 *             position-independent denominator of utils.vector_intercepts (utils.py:96) for
 *             ray k against wall w (inf for parallel pairs, like the zero-jitter reference)
 *  cells      device float32 [4][n]: row 0 = a*mu_d, row 1 = a with
 *             a = sqrt(log2(e)/2)/sigma_d (so the radial gaussian is
 *             exp2(-(a d - a mu_d)^2)), row 2 = kappa*log2(e), row 3 unused
 *  vm_table   device float32, rows padded to Kp = K rounded up to a multiple of 4.
 *             allocentric: [n][Kp] = log2(e)*kappa_c*(cos(theta_k - mu_c) - 1).
 *             egocentric: [2][n][Kp] = cos(theta_k - mu_c), sin(theta_k - mu_c).
 *             Pad entries (k >= K): any finite value or -inf (the kernel gives the
 *             pad directions an infinite distance, so their terms vanish); the head bearing
 *             utils.get_angle(head_direction) enters through its cosine / sine
 *             (hx+1e-6, hy)/norm, so no per-term trigonometry is needed
 *  inv_norm   device float32 [n]: 1 / cell_fr_norm (Neurons.py:1598-1604)
 *  egocentric needs io->hd_x / hd_y
 *  ray_out    device float32 [T][K][B] or NULL: the first-wall ray distances
 *             (diagnostic / parity of the ray stage) */
int riab_boundary_vector_cells(const RiabEnv* env, const RiabRateIO* io, const double* test_dirs,
                               const double* ray_rden, int32_t K, const float* cells, const float* vm_table,
                               const float* inv_norm, int32_t n, int32_t egocentric,
                               float* ray_out, riab_stream_t stream);

/* riab_boundary_vector_cells with direction windows for allocentric cells (K % 4 == 0): the caller hands
 * the tables (table, vm_table, inv_norm) in an order of its choosing — cell_rows[i] = the cell (output
 * row) table row i belongs to — and, per group g of four consecutive table rows, windows[2g] = first test
 * direction and windows[2g+1] = number of directions (both multiples of 4; the window wraps modulo K) to
 * accumulate; directions outside the window are skipped.  The caller chooses the windows so that every
 * skipped von Mises weight is negligible (ratinabox_amd: below 2^-24 of the peak).  cell_rows / windows:
 * device int32 [n] / [ceil(n/4)][2]; both NULL = riab_boundary_vector_cells. */
int riab_boundary_vector_cells_windowed(const RiabEnv* env, const RiabRateIO* io, const double* test_dirs,
                                        const double* ray_rden, int32_t K, const float* table,
                                        const float* vm_table, const float* inv_norm, int32_t n,
                                        int32_t egocentric, float* ray_out, const int32_t* cell_rows,
                                        const int32_t* windows, riab_stream_t stream);

/* ObjectVectorCells.get_state (Neurons.py:1991-2116; FieldOfViewOVCs Neurons.py:2119-2150 is
 * the same on a radial manifold) with Environment.get_distances_between___accounting_for_
 * environment(..., return_vectors=True) (Environment.py:677-730).
 *  objects       device float32 [n_objects][2] (Environment.objects["objects"])
 *  object_types  device int32 [n_objects]
 *  cells         device float32 [n][6] = (a*mu_d, a, cos(mu_theta), sin(mu_theta), kappa*log2(e),
 *                tuning type), a = sqrt(log2(e)/2)/sigma_d
 *  walls_occlude line_of_sight geometry: an object behind an internal wall (env->walls[4:]) is
 *                at distance 1000 (Neurons.py:1938-1941); egocentric needs io->hd_x / hd_y */
int riab_object_vector_cells(const RiabEnv* env, const RiabRateIO* io, const float* objects,
                             const int32_t* object_types, int32_t n_objects, const float* cells, int32_t n,
                             int32_t walls_occlude, int32_t egocentric, riab_stream_t stream);

/* AgentVectorCells.get_state (+ FieldOfViewAVCs), Neurons.py:2204-2355: ObjectVectorCells whose one
 * object is ANOTHER AGENT, i.e. a different point for every lane: other_x / other_y are float32 rows
 * laid out like io->pos_x / pos_y (element [t*other_ld + b]; other_ld = 0 repeats one row for every t).
 * cells as for riab_object_vector_cells (the type column is ignored: every cell responds to the agent). */
int riab_agent_vector_cells(const RiabEnv* env, const RiabRateIO* io, const float* other_x, const float* other_y,
                            int64_t other_ld, const float* cells, int32_t n, int32_t walls_occlude,
                            int32_t egocentric, riab_stream_t stream);

/* Neurons.save_to_history spike rule (Neurons.py:681-687) on rates that already
 * exist: spikes = u < dt*rate (one fp32 multiply, one fp32 compare).  count
 * elements, u_in explicit uniforms or NULL => Philox as in the rate kernels
 * (then n, B, T describe the [T][n][B] shape). */
int riab_spikes(const RiabRateIO* io, int32_t n, riab_stream_t stream);

/* Neurons.update noise (Neurons.py:153-168) for T consecutive steps: for each time row t,
 * noise += OU(noise; 0, noise_std, noise_coherence_time) and rates[t][c][b] += noise[c][b].
 * noise device float32 [n][B] read/written; rates device float32 [T][n][B]; z_in device
 * float32 [T][n][B] standard normals or NULL => Philox keyed by (seed; step+t, cell, agent).
 * theta_dt = dt/tau, sigma_dt = sqrt(2 std^2/(tau dt)) * dt (utils.py:361-368). */
int riab_neuron_noise(float* noise, float* rates, const float* z_in, int32_t n, int64_t B, int32_t T,
                      float theta_dt, float sigma_dt, uint64_t seed, uint64_t step, int32_t pop_id,
                      int64_t agent_id0, riab_stream_t stream);

/* One input layer of a FeedForwardLayer: its firing rates and its weights. */
typedef struct RiabFFInput {
  const float* rates;  /* device float32 [T][n_in][B]: the input layer's rates for the same rows */
  const float* wt;     /* device float32 [n_in][Mp]: the weight matrix TRANSPOSED (w[m][k] -> wt[k][m]),
                          rows padded with zeros to Mp = n_out rounded up to a multiple of 32 */
  int32_t n_in;
} RiabFFInput;

enum { RIAB_ACT_LINEAR = 0, RIAB_ACT_SIGMOID = 1, RIAB_ACT_RELU = 2, RIAB_ACT_TANH = 3,
       RIAB_ACT_RETANH = 4, RIAB_ACT_SOFTMAX = 5 };

/* FeedForwardLayer.get_state (Neurons.py:2797-2847) with utils.activate (utils.py:919-1026):
 * out[t][m][b] = act(sum_l sum_k w_l[m][k] * rates_l[t][k][b] + bias[m]) on the fp32 matrix
 * cores (v_mfma_f32_32x32x2_f32, exact fp32 accumulation).  Up to 8 input layers.
 * act_params[4]: sigmoid (max_fr, min_fr, mid_x, beta = ln(19)/(width_x/2)); relu / tanh /
 * retanh / softmax (gain, threshold, -, -); linear ignores them.
 * out_prime (or NULL) receives the derivative of the activation at the same pre-activation
 * (FeedForwardLayer.firingrate_prime, Neurons.py:2839-2845). */
int riab_feedforward(const RiabFFInput* inputs, int32_t n_inputs, const float* bias, int32_t n_out,
                     int64_t T, int64_t B, int32_t activation, const float* act_params, float* out,
                     float* out_prime, riab_stream_t stream);

/* ---- step plans: the closed-loop per-step path as one native call -----------------------------
 * `for each step: Agent.update(); N.update() for N in Agent.Neurons` (demos/simple_example.ipynb
 * cell 4; contribs/TaskEnvironment.py:399-408) recorded once, then advanced by riab_plan_step:
 * row cursors, RNG counters and pointers are kept in C++, every kernel of every step is enqueued on
 * `stream`, nothing is allocated or synchronised. */
enum { RIAB_POP_PLACE = 0, RIAB_POP_GRID = 1, RIAB_POP_HDC = 2, RIAB_POP_BVC = 3, RIAB_POP_OVC = 4, RIAB_POP_FF = 5,
       RIAB_POP_VELOCITY = 6, RIAB_POP_SPEED = 7, RIAB_POP_RANDOM_SPATIAL = 8 };
#define RIAB_FF_MAX_INPUTS 8

typedef struct RiabPopulation {
  int32_t kind;              /* RIAB_POP_* */
  int32_t n;                 /* cells */
  RiabRateIO io;             /* min_fr, max_fr, pop_id are used; pointers / steps are filled per step */
  float* rates_base;         /* device float32 [capacity_rows][n][B] history chunk */
  uint8_t* spikes_base;      /* device uint8 [capacity_rows][n][B] or NULL */
  int64_t capacity_rows;     /* 0: rates_base is a single row overwritten every step */
  const float* table;        /* the population's cell table (as in its own entry point) */
  int32_t description;       /* place / grid */
  int32_t geometry;          /* place */
  float top_hat_width;       /* place */
  float f0;                  /* grid */
  const double* test_dirs;   /* bvc */
  const double* ray_rden;    /* bvc */
  int32_t K;                 /* bvc */
  int32_t egocentric;        /* bvc / ovc */
  const float* vm_table;     /* bvc */
  const float* inv_norm;     /* bvc */
  const int32_t* cell_rows;  /* bvc direction windows (riab_boundary_vector_cells_windowed) or NULL */
  const int32_t* windows;    /* bvc */
  /* bvc, step plans only, optional (NULL: none): scratch for the ray exchange of a one-row launch — the workgroups that
   * share a tile of 64 positions each cast a share of its K rays and read the others' (csrc/riab_bvc.hip).  bvc_xch:
   * device float32 [B / 64 rounded up][K rounded up to 4][64]; bvc_xch_count: device uint32 [B / 64 rounded up], ZEROED by
   * the caller when the plan is created (the plan counts its launches against it) */
  float* bvc_xch;
  uint32_t* bvc_xch_count;
  const float* objects;      /* ovc */
  const int32_t* object_types; /* ovc */
  int32_t n_objects;         /* ovc */
  int32_t walls_occlude;     /* ovc */
  float one_sigma_speed;     /* velocity / speed */
  const float* targets;      /* random spatial: [n_anchors][n]; `table` holds the anchors */
  int32_t n_anchors;         /* random spatial */
  /* additive OU noise of Neurons.update (Neurons.py:153-168, riab_neuron_noise); NULL = none */
  float* noise_state;        /* device float32 [n][B] */
  float noise_theta_dt;      /* dt / noise_coherence_time */
  float noise_sigma_dt;      /* sqrt(2 std^2 / (tau dt)) * dt */
  /* RIAB_POP_FF (FeedForwardLayer, riab_feedforward): the inputs are populations added to the plan
   * BEFORE this one; each step reads the rows they have just written */
  int32_t n_inputs;
  int32_t input_index[RIAB_FF_MAX_INPUTS];      /* plan indices of the input populations */
  const float* input_wt[RIAB_FF_MAX_INPUTS];    /* their W^T tables (RiabFFInput.wt) */
  const float* bias;         /* [n] */
  int32_t activation;        /* RIAB_ACT_* */
  float act_params[4];
  float* rates_prime;        /* device float32 [n][B] activation derivative, or NULL */
} RiabPopulation;

typedef struct RiabPlan RiabPlan;

/* state / diag / row_scratch as in riab_agent_step (row_scratch: device float32 [8][B], receives the
 * newest history row when no agent history chunk is attached); step = Agent updates taken so far. */
RiabPlan* riab_plan_create(const RiabEnv* env, const RiabMotion* motion, double* state, int64_t B,
                           int64_t agent_id0, uint64_t seed, uint64_t step, float* row_scratch, int32_t* diag);
void riab_plan_destroy(RiabPlan* plan);
int riab_plan_set_motion(RiabPlan* plan, const RiabMotion* motion, const double* drift);
/* imported / forced trajectories (Agent.py:229-266): the coming `n_rows` agent steps of the plan move the agents to
 * the rows of `forced` (device float64 [n_rows][2][B]) instead of running the motion model, exactly as
 * riab_agent_step(forced_pos = row) would; RIAB_EFULL when they are used up.  NULL: back to the motion model. */
int riab_plan_set_forced(RiabPlan* plan, const double* forced, int64_t n_rows);
int riab_plan_set_agent_history(RiabPlan* plan, float* hist_base, int64_t capacity_rows);
int riab_plan_add(RiabPlan* plan, const RiabPopulation* pop);  /* returns the population's index */
int riab_plan_set_population_history(RiabPlan* plan, int32_t index, float* rates_base, uint8_t* spikes_base,
                                     int64_t capacity_rows);
int64_t riab_plan_rows_free(const RiabPlan* plan);
uint64_t riab_plan_step_index(const RiabPlan* plan);
/* n_steps x (Agent.update(); every population's update()); RIAB_EFULL (nothing launched) when a
 * history chunk has fewer than n_steps free rows. */
int riab_plan_step(RiabPlan* plan, int32_t n_steps, riab_stream_t stream);
/* The two halves of one plan step as separate calls — Agent.update() / one population's Neurons.update() on the
 * agent's newest history row — for callers that keep the reference's per-object call structure
 * (demos/simple_example.ipynb cell 4: `Ag.update(); PCs.update()`): same kernels, arguments and RNG counters as
 * riab_plan_step.  RIAB_EFULL when the object's history chunk is exhausted (attach a new one).  Not with a task. */
int riab_plan_step_agent(RiabPlan* plan, riab_stream_t stream);
int riab_plan_step_population(RiabPlan* plan, int32_t index, riab_stream_t stream);
/* The row riab_plan_step_agent wrote ahead for the fused population is stale — something that population reads was
 * edited between the two calls of the step (contribs/TaskEnvironment.py:323-330: a reset teleports agents and patches
 * agent.history["pos"][-1]): it is not claimed; riab_plan_step_population launches the population's own kernel on the
 * edited row instead. */
int riab_plan_discard_ahead(RiabPlan* plan);

/* The closed-loop step in ONE launch (csrc/riab_step1.hip).  Replaces, per step, the reference's
 * `Agent.update()` (Agent.py:160-242) + `Neurons.update()` (Neurons.py:145-171) of ONE population — the pair
 * contribs/TaskEnvironment.py:399-408 and every user loop call per step — which riab_plan_step otherwise issues as two
 * dependent launches (riab_agent_step(T = 1), then the population's kernel).  Every workgroup of the launch advances the
 * 256 agents whose rates it then writes; the workgroups of a 256-agent segment read the same float64 state and ONE of
 * them writes it back, after the others have reported in `sync_words` that their loads have returned.  Bit-identical
 * to the two-launch step.  No allocation, no synchronisation, nothing process-wide: capturable like every other entry
 * point.
 *   sync_words  device uint32 [RIAB_STEP1_SYNC_WORDS(B)], zeroed by the caller once, the plan's for its lifetime (NULL:
 *               switch the one-launch step off).  Layout: RIAB_STEP1_SYNC_STRIDE words per 256-agent segment — word y
 *               (1 <= y < RIAB_STEP1_SYNC_MAX_Y) = arrival word of the segment's workgroup y, holding the epoch of the
 *               plan's last one-launch step —, then RIAB_STEP1_SYNC_TAIL counters: [RIAB_STEP1_SYNC_TIMEOUTS] writers
 *               that gave up waiting (~1 s; must stay 0: the state was then written while a workgroup of the grid had
 *               not run yet), then float64 [RIAB_MAX_WALLS][6]: the wall table as the kernels keep it (start, direction,
 *               1 / |direction|^2, 1 / |direction|) and one word (+ padding) saying whether the first four walls are the
 *               edges of a solid rectangular room (the motion step's box fast path), both prepared by the plan's first
 *               one-launch step (a one-wave kernel in front of it) and again when the repel distance changes; then
 *               RIAB_STEP1_MAIL_STRIDE words per segment for a plan with a task attached: what the segment's writer
 *               tells its other workgroups about this step's resets (which of its agents a reset moved, and where to, as
 *               the history row keeps positions).
 * What is fused: motion steps (Philox noise, drift or not; no forced trajectory) of whole 256-agent segments, with the
 * plan's LARGEST population among PlaceCells (euclidean geometry, not one_hot), GridCells and HeadDirectionCells
 * without additive noise; the other populations follow as their own kernels in list order, as before.  A plan with a
 * task attached (riab_plan_set_task; RIAB_OPT_FUSED_TASK on) gets the rest of TaskEnvironment.step, the auto-reset and
 * the next scripted action in the same launch as well (riab_plan_step only): the segment's writer workgroup keeps the
 * task's books, the others learn from the mail which agents a reset moved and write those agents' rates again.  With the split entry points riab_plan_step_agent launches the fused kernel (the population's row is written
 * ahead of its update() call) and the population's riab_plan_step_population of the same step only advances the row
 * cursor; a plan whose populations are not updated after each agent step stops fusing.
 * riab_plan_info: 0 steps served by the one-launch kernel, 1 index of the first fused population (-1: none), 2 kernels
 * launched by the plan so far, 3 non-zero when arrival words are attached, 4 how many populations ride in the launch,
 * 5 the compute units the plan counts on, 8 .. 8 + RIAB_STEP1_MAX_POPS - 1 their indices (-1: unused). */
#define RIAB_STEP1_SYNC_STRIDE 64
#define RIAB_STEP1_SYNC_MAX_Y 64     /* workgroups per segment (arrival words 1 .. 63) */
#define RIAB_STEP1_SYNC_TAIL 16
#define RIAB_STEP1_SYNC_TIMEOUTS 0
#define RIAB_STEP1_SYNC_FIRST_BAD 1  /* the first / last step a wait gave up in, as "agent steps taken once it was done" (0: none) */
#define RIAB_STEP1_SYNC_LAST_BAD 2
#define RIAB_STEP1_SYNC_FATAL 3      /* give-ups after which the STATE is not trustworthy (one-world task plans only: a writer that never saw the world's verdict) */
#define RIAB_STEP1_SYNC_WALLS_AT(B) ((((B) + 255) / 256) * RIAB_STEP1_SYNC_STRIDE + RIAB_STEP1_SYNC_TAIL)
#define RIAB_STEP1_SYNC_MAIL_AT(B) (RIAB_STEP1_SYNC_WALLS_AT(B) + 12 * RIAB_MAX_WALLS + 4)
#define RIAB_STEP1_MAIL_STRIDE 1088  /* per segment, 8-byte entries (epoch << 32 | value): (at 0) 4 x 8 verdict entries = per mover wave its lane mask in halves and its first two movers' x, y; (at 64) x[256], y[256] */
#define RIAB_STEP1_SYNC_WORDS(B) (RIAB_STEP1_SYNC_MAIL_AT(B) + (((B) + 255) / 256) * RIAB_STEP1_MAIL_STRIDE)
int riab_plan_set_fused(RiabPlan* plan, uint32_t* sync_words, int64_t n_words);
int64_t riab_plan_info(const RiabPlan* plan, int32_t which);
/* How much of the device the plan's launches can occupy.  The one-launch step cuts its grid to ONE round of workgroups,
 * and a task plan's form of it — whose workgroups wait for each other in both directions — is only used when that whole
 * grid is resident at once: compute units x workgroups of the kernel per unit (asked of the runtime per instantiation:
 * hipOccupancyMaxActiveBlocksPerMultiprocessor); otherwise the plan keeps its two launches per step.  n_cus = 0
 * (default): hipDeviceAttributeMultiprocessorCount of the calling thread's device.  A caller that knows better says so:
 * a process whose queues carry a CU mask (HSA_CU_MASK, ROC_GLOBAL_CU_MASK, hipExtStreamCreateWithCUMask) is reported
 * the whole device by the runtime — riab_probe_compute_units counts what its workgroups really land on. */
int riab_plan_set_compute_units(RiabPlan* plan, int32_t n_cus);
/* Count the compute units the workgroups of `stream` land on: two launches on `stream` — every one-wave workgroup of a
 * grid several times the device marks its unit (XCC_ID, HW_ID's SE / SH / CU fields) in scratch, a one-workgroup kernel
 * counts the marks into scratch[RIAB_CU_PROBE_WORDS - 1].  scratch: device uint32 [RIAB_CU_PROBE_WORDS], ZEROED by the
 * caller; no allocation, no synchronisation, no copy: the caller reads the last word once the stream has got there. */
#define RIAB_CU_PROBE_WORDS 4097
int riab_probe_compute_units(uint32_t* scratch, riab_stream_t stream);

/* ---- batched TaskEnvironment (contribs/TaskEnvironment.py) ------------------------------------
 * The closed-loop caller of the path: `TaskEnvironment.step(actions)` = Agent.update(drift_velocity
 * = action) [riab_agent_step / riab_plan_step], then the task bookkeeping of
 * contribs/TaskEnvironment.py:410-449: reward caches decay (RewardCache.update :913-927,
 * Reward.update :804-821), the clock advances (:358), goals are checked and consumed
 * (GoalCache.check :1076-1152, SpatialGoal.check :1337-1360 with the line-of-sight distance of
 * Environment.py:677-779, TimeElapsedGoal.check :1271-1278), the termination delay is appended
 * (:420-434) and rewards are totalled (RewardCache.get_total :929-939).
 *
 * Batched semantics: every lane (agent of the batch) is an independent single-agent replica of the
 * reference's TaskEnvironment sharing one goal pool and one clock; lanes shard over GPUs like agents
 * do.  riab_task_step is that bookkeeping for all lanes in one launch; riab_task_reset is
 * TaskEnvironment.reset (:307-351, GoalCache.reset :1218-1252) for the lanes selected by a mask.
 *
 * Per-lane task state: device float64 [RIAB_TS_ROWS][B], lane axis fastest (small integers are
 * stored as doubles so that the whole state is one tensor). */
#define RIAB_TASK_MAX_GOALS 16   /* goals in one episode's list, incl. the termination-delay goal */
#define RIAB_TASK_MAX_REWARDS 32 /* simultaneously active rewards of one lane */
#define RIAB_TASK_MAX_POOL 64    /* goals in the pool the episodes draw from */
enum {
  RIAB_TS_N_GOALS = 0,        /* length of the lane's goal list */
  RIAB_TS_DELAYED = 1,        /* episode_state["delayed_term"] */
  RIAB_TS_PAD_START = 2,      /* TimeElapsedGoal.start_time of the termination-delay goal */
  RIAB_TS_N_REWARDS = 3,      /* length of the lane's RewardCache.cache */
  RIAB_TS_EPISODE = 4,        /* TaskEnvironment.episode */
  RIAB_TS_EP_START = 5,       /* episodes["start"][-1] */
  RIAB_TS_EP_ANY_ENDED = 6,   /* an episode of non-zero duration has ended before */
  RIAB_TS_STEPS_ACTIVE = 7,   /* RewardCache.stats total_steps_active / total_steps_inactive, max, min */
  RIAB_TS_STEPS_INACTIVE = 8,
  RIAB_TS_R_MAX = 9,
  RIAB_TS_R_MIN = 10,
  RIAB_TS_STARTED = 11,       /* len(episodes["start"]) > 0 */
  RIAB_TS_GOAL_LIST = 12,     /* [MAX_GOALS] pool index of list entry i; RIAB_GOAL_TIME_ELAPSED = delay goal */
  RIAB_TS_RW_STATE = RIAB_TS_GOAL_LIST + RIAB_TASK_MAX_GOALS,     /* [MAX_REWARDS] Reward.state, in cache (append) order */
  RIAB_TS_RW_EXPIRE = RIAB_TS_RW_STATE + RIAB_TASK_MAX_REWARDS,  /* [MAX_REWARDS] Reward.expire_clock */
  RIAB_TS_RW_SRC = RIAB_TS_RW_EXPIRE + RIAB_TASK_MAX_REWARDS,    /* [MAX_REWARDS] pool index of the goal that gave it */
  RIAB_TS_ROWS = RIAB_TS_RW_SRC + RIAB_TASK_MAX_REWARDS          /* = 124 */
};
#define RIAB_GOAL_TIME_ELAPSED (-2)
/* goal pool row (float64 x 8): x, y, radius, then the goal's Reward: init_state, dt, expire_clock,
 * decay preset (RIAB_DECAY_*), decay knob (Reward.decay_preset / decay_knobs_preset, :732-743) */
#define RIAB_GOAL_COLS 8
enum { RIAB_DECAY_CONSTANT = 0, RIAB_DECAY_LINEAR = 1, RIAB_DECAY_EXPONENTIAL = 2, RIAB_DECAY_NONE = 3 };
enum { RIAB_GOALORDER_NONSEQUENTIAL = 0, RIAB_GOALORDER_SEQUENTIAL = 1 };
/* task diagnostics (int32 counters, device, [4]) */
enum { RIAB_TD_REWARD_OVERFLOW = 0, RIAB_TD_LATE_COMPLETIONS = 1, RIAB_TD_EPLOG_OVERFLOW = 2, RIAB_TD_RESETS = 3 };

typedef struct RiabTask {
  const double* goals;         /* device float64 [n_pool][RIAB_GOAL_COLS] */
  int32_t n_pool;
  int32_t goalorder;           /* RIAB_GOALORDER_* (GoalCache.goalorder) */
  double terminate_delay;      /* episode_terminate_delay (seconds; 0 = none) */
  double pad_reward[5];        /* no_reward_default (:949-951): init_state, dt, expire_clock, preset, knob */
  double default_reward_level; /* RewardCache.default_reward_level */
} RiabTask;

/* One TaskEnvironment.step after the agents have moved.  pos_x / pos_y: device float64 [B] (rows
 * RIAB_S_POS_X / RIAB_S_POS_Y of the agent state); t_env: the clock AFTER `self.t += self.dt`.
 * reward_out float64 [B] = RewardCache.get_total(); terminal_out uint8 [B].  Each lane runs the
 * goal check passes of one reference step (two, three when the termination delay is appended) with
 * the reference's list semantics (index skipped after a pop, :1130-1141; reward skipped after an
 * expiry, :919-922).  terminal_out is "no goals left" after the last pass; the reference returns
 * the value after the first (it differs only when a later pass of the same step completes further
 * goals — counted in diag[RIAB_TD_LATE_COMPLETIONS]; the reference then errors on its next step). */
int riab_task_step(const RiabEnv* env, const RiabTask* task, double* task_state, const double* pos_x,
                   const double* pos_y, int64_t B, double t_env, double* reward_out, uint8_t* terminal_out,
                   int32_t* diag, riab_stream_t stream);

/* get_goal_vector (contribs/TaskEnvironment.py:1555-1584): out = goal - position for the head of the
 * lane's list (sequential) or its nearest pending spatial goal, (0,0) when none is pending.
 * scale > 0 writes scale * unit vector instead (0 where there is no goal): the scripted
 * goal-seeking action of the reference's test loop (:1599-1605). */
int riab_task_goal_vector(const RiabEnv* env, const RiabTask* task, double* task_state, const double* pos_x,
                          const double* pos_y, int64_t B, double scale, double* out_x, double* out_y,
                          riab_stream_t stream);

/* TaskEnvironment.reset for the lanes with mask[b] != 0 (mask NULL = every lane): the running
 * episode is closed (appended to ep_log when given: float64 [cap][5] = global lane id, episode,
 * start, end, duration; ep_count int32 device counter), the episode counter advances unless the
 * closed episode had zero duration (:333-337), the goal list is refilled with n_select goals of
 * the pool — the first n_select when `ordered` (reset_orders_goal), otherwise a uniform sample
 * without replacement from Philox(seed; counter, global lane id) — and, when `teleport`, the
 * lane's position (pos_x / pos_y: rows RIAB_S_POS_X / _Y of the agent state; hist_x / hist_y: rows
 * RIAB_H_POS_X / _Y of the newest float32 history record, or NULL) is set to (new_x, new_y)[b] or,
 * when those are NULL, drawn as Environment.sample_positions(1) does for a square box: centre +
 * U(-0.45, 0.45) * scale per axis (Environment.py:560-633).  All per-lane arrays hold B entries. */
int riab_task_reset(const RiabEnv* env, const RiabTask* task, double* task_state, const uint8_t* mask, int64_t B,
                    int64_t agent_id0, double t_env, int32_t n_select, int32_t ordered, uint64_t seed,
                    uint64_t counter, int32_t teleport, const double* new_x, const double* new_y, double* pos_x,
                    double* pos_y, float* hist_x, float* hist_y, double* ep_log, int64_t ep_log_cap,
                    int32_t* ep_count, int32_t* diag, riab_stream_t stream);

/* ---- the other batching of the task: the lanes are the agents of ONE world ----------------------------
 * The reference's TaskEnvironment with several Agents and agentmode = "interact" (its default,
 * contribs/TaskEnvironment.py:1030): one clock, one episode, and — because GoalCache.pop removes a satisfied
 * goal from every agent's list (:1165-1172) while reset / append fill all lists alike (:1204-1211, :1250-1252)
 * — ONE shared goal list; every agent keeps its own reward cache (rows RIAB_TS_N_REWARDS, RIAB_TS_STEPS_*,
 * RIAB_TS_R_*, RIAB_TS_RW_* of task_state; its goal / episode rows are unused).  GoalCache.check (:1076-1152)
 * gives the agents their turns in agent order against the list as the earlier agents of the pass left it: a
 * goal goes to the FIRST agent, in that order, found inside it; in a nonsequential turn the goal that slides
 * into a popped slot is skipped by that agent and left to the later ones (:1141); in sequential order every
 * agent looks at the head once per pass, so several heads can go in one pass.
 * Shared state: device float64 [RIAB_TW_ROWS]. */
enum {
  RIAB_TW_N_GOALS = 0,      /* length of the shared goal list */
  RIAB_TW_DELAYED = 1,      /* episode_state["delayed_term"] */
  RIAB_TW_PAD_START = 2,    /* start_time of the termination-delay goal */
  RIAB_TW_EPISODE = 3,      /* TaskEnvironment.episode */
  RIAB_TW_EP_START = 4,     /* episodes["start"][-1] */
  RIAB_TW_EP_ANY_ENDED = 5, /* an episode of non-zero duration has ended before */
  RIAB_TW_STARTED = 6,      /* len(episodes["start"]) > 0 */
  RIAB_TW_TERMINAL = 7,     /* the flag the last step left in terminal_out */
  RIAB_TW_GOAL_LIST = 8,    /* [RIAB_TASK_MAX_GOALS] pool index of list entry i; RIAB_GOAL_TIME_ELAPSED = delay goal */
  RIAB_TW_ROWS = 24
};
/* One TaskEnvironment.step (:410-449) of the world after the agents have moved; arguments as riab_task_step.
 * One launch.  Every lane decays its own rewards and works out which goals of the list it stands in; a lane that
 * stands in none totals its rewards and is done, the others put themselves on a work list (cand_scratch: device
 * int32 [B]; met_scratch: device uint64 [B]).  The workgroup that finishes last (ctl: device int32 [2] — ticket,
 * work-list length — zero before the first call; the kernel leaves them zero) walks the step's check passes over
 * the shared list among the listed lanes, appends the awards to the winners' caches in award order and totals
 * those lanes.  terminal_out [B] holds the world's flag for every agent ("no goals left" after the step's last
 * pass, as riab_task_step); the column is rewritten only when the flag changes (RIAB_TW_TERMINAL remembers it):
 * hand in the same array every step.  No workgroup waits for another: capturable, nothing has to be co-resident.
 * One launch at a time per world (the scratch and ctl belong to the launch in flight): stream-ordered calls only. */
int riab_task_world_step(const RiabEnv* env, const RiabTask* task, double* task_state, double* world,
                         const double* pos_x, const double* pos_y, int64_t B, double t_env, double* reward_out,
                         uint8_t* terminal_out, uint64_t* met_scratch, int32_t* cand_scratch, int32_t* ctl,
                         int32_t* diag, riab_stream_t stream);
/* TaskEnvironment.reset (:307-351) of the world: the episode is closed once (ep_log row: lane id -1), the
 * shared list refilled with n_select goals of the pool (the first ones when `ordered`, else a sample drawn
 * from Philox(seed; counter, id 0xFFFFFFFF)), every agent teleported when `teleport` (to (new_x, new_y)[b] or
 * its own draw, as riab_task_reset).  No mask: there is one episode.  only_if_terminal != 0: the caller's
 * `if terminal: env.reset()` decided on the device — nothing is reset unless the last riab_task_world_step left
 * the world's flag set.  gv_x / gv_y (or NULL): every agent's goal vector AFTER the reset-or-not, as
 * riab_task_world_goal_vector(scale = gv_scale) would write it — a step plan's next scripted action in the same
 * launch (needs pos_x / pos_y). */
int riab_task_world_reset(const RiabEnv* env, const RiabTask* task, double* task_state, double* world, int64_t B,
                          int64_t agent_id0, double t_env, int32_t n_select, int32_t ordered, uint64_t seed,
                          uint64_t counter, int32_t teleport, const double* new_x, const double* new_y,
                          double* pos_x, double* pos_y, float* hist_x, float* hist_y, double* ep_log,
                          int64_t ep_log_cap, int32_t* ep_count, int32_t only_if_terminal, double gv_scale,
                          double* gv_x, double* gv_y, int32_t* diag, riab_stream_t stream);
/* get_goal_vector (:1555-1584) of every agent against the shared list (arguments as riab_task_goal_vector). */
int riab_task_world_goal_vector(const RiabEnv* env, const RiabTask* task, double* task_state, const double* world,
                                const double* pos_x, const double* pos_y, int64_t B, double scale, double* out_x,
                                double* out_y, riab_stream_t stream);

/* Attach a task (riab_task_*) to the plan: every plan step then is one TaskEnvironment.step —
 * [scripted_speed > 0: drift <- scripted_speed * unit goal vector, written to the plan's drift buffer]
 * Agent.update, clock += dt_env, riab_task_step, [auto_reset: riab_task_reset(mask = terminal_out) with
 * the reset counter advanced every step], then the populations (which therefore see teleported
 * positions).  Task arrays hold task_B lanes (the agents of the batch, without the padding of the
 * plan's B).  task == NULL detaches. */
int riab_plan_set_task(RiabPlan* plan, const RiabTask* task, double* task_state, int64_t task_B, double t_env,
                       double dt_env, double* reward_out, uint8_t* terminal_out, int32_t* task_diag,
                       int32_t auto_reset, int32_t n_select, int32_t ordered, uint64_t task_seed,
                       uint64_t reset_counter, int32_t teleport, double* ep_log, int64_t ep_log_cap,
                       int32_t* ep_count, double scripted_speed);
double riab_plan_task_clock(const RiabPlan* plan);
/* ... of the kind whose lanes are the agents of ONE world (riab_task_world_*; call after riab_plan_set_task with the
 * same task): a plan step then is [scripted_speed > 0: the goal vectors into the drift buffer — a launch of its own
 * on the first step and when auto_reset is off], Agent.update + clock += dt_env + riab_task_world_step (ONE launch),
 * [auto_reset: riab_task_world_reset(only_if_terminal = 1, with the next step's scripted action), the reset counter
 * advanced every step], then the populations.  world == NULL: back to per-lane. */
int riab_plan_set_task_world(RiabPlan* plan, double* world, uint64_t* met_scratch, int32_t* cand_scratch, int32_t* ctl);

/* ---- the open-loop path as ONE native call: flag-coupled trajectory + firing-rate kernels -------
 * `for t in range(T): Agent.update(); [N.update() for N in Neurons]` (demos/simple_example.ipynb cell 4;
 * Agent.update: ratinabox/Agent.py:160-242, Neurons.update: ratinabox/Neurons.py:145-171) with no kernel boundary
 * between the two stages: the trajectory kernel (float64, one agent's step spread over four specialised waves,
 * csrc/riab_traj4_kernel.h) writes its history rows through to memory and publishes a per-workgroup "steps done"
 * word; the firing-rate stage runs concurrently and consumes the rows as they appear.  Results are bit-identical to
 * T calls of riab_agent_step(T = 1) each followed by the populations' own entry points.
 *
 * Forms of the rate stage, chosen by the call (riab_streamer_last_form):
 *  - a "lead" population — kind RIAB_POP_PLACE (not one_hot) / RIAB_POP_GRID / RIAB_POP_HDC without OU noise, B a
 *    multiple of 256, T <= POLL_MAX — is served by ONE kernel for all its rows whose waves each wait until the 256 agents
 *    of the wave have been stepped past their row (runs of more than 2048 rows: the first HEAD_ROWS rows; the others by
 *    the population's ordinary kernel, 512 rows per launch behind a progress gate);
 *      - a single population: that is the whole stage (RIAB_FORM_ONE_KERNEL / RIAB_FORM_HEAD_AND_PIECES);
 *      - several populations: the largest such population leads if its stores of a row keep pace with the trajectory
 *        kernel (its bytes at 6.5 TB/s against 1.35 us per step + 0.375 us per wall beyond four); the others
 *        (boundary / object vector cells, random spatial neurons, speed cell, FeedForwardLayers — input_index refers
 *        to EARLIER entries of `pops` —, populations with OU noise, further store-bound ones) follow as their ordinary
 *        kernels over the whole run, in array order (RIAB_FORM_POPULATIONS);
 *  - anything else (no lead, runs of more than POLL_MAX rows): per chunk of rows (16, 28, 44, ... 128) a one-wave gate
 *    that waits for the chunk's last row, then each population's ordinary kernel in array order (noise pass and spikes
 *    after it, as in riab_plan_step) (RIAB_FORM_CHUNKS).
 * n_pops == 0 (an agent without populations): the trajectory kernel alone, on `stream`.
 * forced_pos != NULL (Agent.import_trajectory / forced_next_position, Agent.py:229-266): there is no recurrence to
 * overlap: the forced-position kernel and the populations' kernels follow each other on `stream`.
 * Velocity cells (they read the float64 state, which no history row keeps) are not covered: RIAB_EUNSUPPORTED,
 * nothing launched; callers advance them through a step plan.  A `stream` that is being captured into a graph:
 * RIAB_EUNSUPPORTED too (the call queries the stream, and two kernels coupled through words in memory on two streams
 * are not something a graph replay can reproduce).
 *
 * A RiabStreamer holds what the coupling needs besides the kernels: a second HIP stream (the trajectory kernel runs
 * there; the rate stage — which ends last — on `stream`, so that a host synchronisation returns as soon after the
 * last kernel as after any single kernel), two events, and the running count of started trajectory workgroups.  The
 * second stream is borrowed from a process-wide pool: one stream per (device, caller's stream), of the device's highest
 * priority, chosen among up to six candidates by TIMING a pair of launches against the caller's stream at the first
 * call that finds that stream idle — streams that share the caller's hardware queue are 30-50 us slower per pair and
 * are set aside (csrc/riab_simulate.hip: side_stream_for; riab_streamer_info 4-6).
 *
 *  ctrl   device uint32 [RIAB_CTRL_WORDS(B)], zeroed ONCE by the caller when it is created (and that zero-fill
 *         complete before the first call: the kernels that read the words run on the streamer's stream too):
 *         [RIAB_CTRL_STARTED] trajectory workgroups that have become resident (all calls; progress words hold
 *         absolute step counts, the started word accumulates),
 *         [RIAB_CTRL_TIMEOUTS] waves that gave up waiting (must stay 0; results are invalid otherwise),
 *         [RIAB_CTRL_ABORT] set with the first timeout: every later wait returns at once,
 *         [RIAB_CTRL_SERIALISED] calls whose two kernels ran one after the other (see "Residency"),
 *         [RIAB_CTRL_STAMPS .. +3] two uint64: device clock (s_memrealtime) at the first wave's start / the last
 *         wave's end of the row-following rate kernel of the last call that ran one (the clock only moves forward, so
 *         the words need no reset between calls: the newest start is stored, the end is a maximum),
 *         [RIAB_CTRL_TRAJ_STAMPS .. +3] the same clock at the start of trajectory workgroup 0 and at its last
 *         publication (every call; what the form selection's one-off measurement reads, see STEP_NS / LEAD_MBPS),
 *         [RIAB_CTRL_PROGRESS_WORD(w)] (uint32)(step0 + steps whose rows trajectory workgroup w (agents 64w ..
 *         64w+63) has published).  The four words of a 256-agent sub-segment share one 128-byte line that no
 *         other sub-segment touches: every wave of the rate kernel reads exactly one such line, and with all of
 *         them in ONE line (first layout) that line — rewritten by 64 workgroups every few microseconds, so
 *         never served from L2 — throttled the rate kernel to 2.3 TB/s while the trajectory kernel ran [MI355X].
 *  pops   n_pops structs, contiguous: kind, table and parameters, io.min_fr / max_fr / pop_id, rates_base
 *         [capacity_rows][n][B] and spikes_base (or NULL) = row 0 of this call; capacity_rows >= T (callers that
 *         stream through fewer rows issue one call per buffer length: inside a call the rate waves are several
 *         time rows apart, so rows of one call must not alias)
 *  hist   device float32 [T][8][B], required (the rate kernels read it in place)
 *  B      a multiple of 4 (256 for the one-kernel form)
 *  noise  explicit standard normals [T][2][B] (rotation OU, speed OU) instead of the in-kernel Philox draws, or NULL
 *  timed_pop >= 0: that population's kernels are timed (riab_streamer_last_rate_ms afterwards): chunk form: HIP
 *         events around every launch, summed; one-kernel form: timing_mode RIAB_TIMING_STAMPS = the device's
 *         constant clock read by the kernel's first and last waves (no host cost), RIAB_TIMING_EVENTS = start / stop
 *         events attached to the launch (hipExtLaunchKernel; ~7 us of host time in front of the kernel)
 *
 * Residency.  Both kernels must be on the chip at once or the rate waves spin for nothing.  The trajectory kernel is
 * launched first.  Two ways of making sure that the rate kernel cannot fill the chip with waiting waves before the
 * kernel they wait for has been placed (it did, with two processes sharing one GPU):
 *  - RIAB_GATE_RESERVED (the default): when `stream` is idle at the call (hipStreamQuery) and the stage is the
 *    row-following kernel of one population, that kernel is launched as workgroups of TWELVE waves, three per SIMD.
 *    Two of them fit a compute unit, a third does not (8 wave slots per SIMD): on EVERY compute unit one wave slot per
 *    SIMD stays free whatever the rate kernel does, and wave slots are the only resource the two kernels compete for
 *    (the euclidean / periodic place, grid and head-direction kernels hold 32-40 registers per lane and no LDS — asked
 *    of the code object, hipFuncGetAttributes: a kernel with more than 48 registers or with LDS of its own is refused
 *    the shape and takes the gate; a trajectory workgroup needs one slot per SIMD, 224 registers, 80 KB of LDS).  The trajectory workgroups can therefore always be placed: residency by construction,
 *    two launches per call.  (Otherwise, RIAB_GATE_RESERVED behaves as RIAB_GATE_ALWAYS.)
 *  - RIAB_GATE_ALWAYS: a one-wave gate kernel in front of the rate stage (four-wave workgroups, 8 waves per SIMD)
 *    holds it back until every trajectory workgroup of this launch has announced itself in ctrl[RIAB_CTRL_STARTED]:
 *    three launches per call.
 *  - RIAB_GATE_WHEN_BUSY drops that gate when `stream` is idle, with nothing in its place: only for callers that own
 *    the device (kept for A/B runs).
 * Every wait is bounded; a wait that gives up sets ctrl[RIAB_CTRL_ABORT] and is counted in ctrl[RIAB_CTRL_TIMEOUTS].
 * A trajectory workgroup sets its progress word to step0 BEFORE it announces itself, so a call whose step0 is not
 * beyond the last call's rows (the same block replayed: torch.ops.riab.simulate_ in a compiled function) never lets a
 * consumer see the earlier run's words: such calls always take the started gate.
 * ctrl[RIAB_CTRL_SERIALISED] counts calls of >= 8 rows whose rate stage found EVERY row already published when its
 * first wave arrived: the two kernels ran one after the other, not side by side (both streams on one hardware queue —
 * GPU_MAX_HW_QUEUES, many other streams in the process): results are right, the call is up to 45 % slower.
 * All argument checks run before the first launch: an argument error has launched nothing; a failure after the
 * trajectory launch returns RIAB_EPARTIAL. */
enum { RIAB_CTRL_STARTED = 0, RIAB_CTRL_TIMEOUTS = 1, RIAB_CTRL_ABORT = 2, RIAB_CTRL_SERIALISED = 3,
       RIAB_CTRL_STAMPS = 8,        /* two uint64: the timed rate kernel's first-wave start / last-wave end */
       RIAB_CTRL_TRAJ_STAMPS = 12,  /* two uint64: trajectory workgroup 0's start / last publication (every call) */
       RIAB_CTRL_PROGRESS = 32 };
#define RIAB_CTRL_PROGRESS_WORD(w) (RIAB_CTRL_PROGRESS + 32 * ((w) >> 2) + ((w) & 3))
#define RIAB_CTRL_WORDS(B) (RIAB_CTRL_PROGRESS + 32 * (((B) + 255) / 256))
enum { RIAB_TIMING_STAMPS = 0, RIAB_TIMING_EVENTS = 1 };
typedef struct RiabSimulate {
  const RiabEnv* env;
  const RiabMotion* motion;
  double* state;              /* [RIAB_STATE_ROWS][B] */
  int64_t B;
  int64_t agent_id0;
  const double* drift;        /* [2][B] or NULL */
  const double* noise;        /* [T][2][B] or NULL */
  const double* forced_pos;   /* [T][2][B] or NULL */
  const double* resample_pos; /* [T][2][B] or NULL: replacement positions of the resample boundary condition */
  uint64_t seed;
  uint64_t step0;
  int32_t T;
  int32_t n_pops;
  const struct RiabPopulation* pops;
  float* hist;
  int32_t* diag;
  uint32_t* ctrl;
  int32_t timed_pop;
  int32_t timing_mode;
  /* Host arrays the caller's cached device tables were built from (cell centres, widths, wall arrays ...), each with
   * the snapshot taken when the tables were built: compared (memcmp) before anything is launched; a difference returns
   * RIAB_ECHANGED.  Lets a caller that re-issues the same run skip its own content checks (users of the Python layer
   * edit tuning arrays in place, reference tests/test_advanced.py:59).  NULL / 0: nothing to compare. */
  const struct RiabWatch* watch;
  int32_t n_watch;
} RiabSimulate;
typedef struct RiabWatch {
  const void* live;
  const void* snapshot;
  int64_t bytes;
} RiabWatch;
typedef struct RiabStreamer RiabStreamer;
RiabStreamer* riab_streamer_create(void);
void riab_streamer_destroy(RiabStreamer* h);
/* options of a streamer: GATE (RIAB_GATE_ALWAYS, the default, or RIAB_GATE_WHEN_BUSY: see "Residency"); POLL_MAX
 * (default 65535, the most the row-following kernel's grid holds; 0: never): longer runs take the chunk form of the
 * rate stage; HEAD_ROWS (default 256): runs of more than 2048 rows give only their first HEAD_ROWS rows to the
 * row-following kernel and the rest to the population's ordinary kernel behind progress gates (65535: never) */
/* ... SIDE_STREAM: where the trajectory kernel runs — 0 (default) a stream of the streamer's own at the device's
 * highest priority, 1 a stream of its own at the default priority, 2 the caller's stream (the two kernels then run one
 * after the other: for tests of the RIAB_CTRL_SERIALISED diagnostic); takes effect at the next call.
 * STEP_NS / LEAD_MBPS: what the choice between the populations form and the chunk form compares — a trajectory step,
 * in nanoseconds, and the lead population's store rate in MB/s: the lead leads when its stores of a row take at least
 * 1.5 steps (2 steps when the step time is a measurement: it was not taken next to a store stream).  0 (default):
 * measured — the first call with several populations that finds `stream` idle reads the device-clock stamps the
 * previous call left in ctrl (one blocking 32-byte copy, once per streamer) and keeps the step time as a factor on the
 * built-in figure; until then, and when the stamps are unusable, the constants measured on MI355X (900 ns + 250 ns per
 * wall beyond four; 6.5 TB/s).  A non-zero value replaces the measurement (tests; chips whose clocks are known to
 * differ). */
/* STRICT (0 / 1): see "Two modes" below.  SPIN_LIMIT: polls before a waiting rate wave / gate gives up (0, the default:
 * 2^20 for waves, 2^22 - 2^24 for gates, about a second; tests set it to 1 to force the abort path). */
enum { RIAB_STREAMER_OPT_GATE = 0, RIAB_STREAMER_OPT_POLL_MAX = 1, RIAB_STREAMER_OPT_HEAD_ROWS = 2,
       RIAB_STREAMER_OPT_SIDE_STREAM = 3, RIAB_STREAMER_OPT_STEP_NS = 4, RIAB_STREAMER_OPT_LEAD_MBPS = 5,
       RIAB_STREAMER_OPT_STRICT = 6, RIAB_STREAMER_OPT_SPIN_LIMIT = 7 };
enum { RIAB_GATE_ALWAYS = 0, RIAB_GATE_WHEN_BUSY = 1, RIAB_GATE_RESERVED = 2 };
int riab_streamer_configure(RiabStreamer* h, int32_t option, int32_t value);
/* Two modes of riab_simulate, against the contract SURVEY.md 8(b2) sets for the boundary ("no allocation, no sync, no
 * host<->device copies inside; enqueue on the passed stream and return; hipGraph-capturable; re-entrant, no globals"):
 *
 *  DEFAULT (what Agent.simulate() and bench.py run: tuned for one short call per synchronisation).  Deviates from that
 *    contract in documented ways, each measured into existence (docs/EXPERIMENTS.md r03 / r04): the trajectory kernel's
 *    stream is taken from a process-wide pool and SCREENED by timing candidate streams the first time the caller's
 *    stream is found idle (synchronises it; ~1 ms, once per (device, stream)); the caller's stream is queried
 *    (hipStreamQuery) at every call; a call with several populations reads the previous call's device-clock stamps
 *    once per streamer (a 32-byte blocking copy); timing events are created on first use.  Not capturable.
 *  STRICT (riab_streamer_configure(RIAB_STREAMER_OPT_STRICT, 1); imposed on any call whose `stream` is being captured):
 *    conforms.  Nothing is allocated, synchronised, queried or copied, nothing outside the streamer is touched: the
 *    trajectory kernel runs on a stream of the streamer's own, the call opens with a one-wave kernel that re-bases the
 *    announcement counter and the call's progress words ON THE DEVICE (so a replay of the captured call finds what the
 *    first run found), the two streams are forked and joined with the streamer's two events at both ends, the started
 *    gate is always taken (graph branches may be serialised by the runtime: the gate makes that safe,
 *    ctrl[RIAB_CTRL_SERIALISED] makes it visible), calls are timed only with events that already exist.  Costs two
 *    more launches and the fork's dependency per call (cfg 2, 20 steps: see DESIGN.md 3.8).  Needs
 *    riab_streamer_warmup(h, stream) once beforehand — outside any capture —, which creates that stream and the events
 *    and also performs the default mode's screening for `stream`; without it a strict call returns RIAB_EUNSUPPORTED
 *    (nothing launched).  Same rows, bit for bit, in either mode.
 *  Which mode a call takes: RIAB_STREAMER_OPT_STRICT = 2 (the default of a new streamer): STRICT for runs of more than
 *    256 steps once riab_streamer_warmup has been called — two launches more are 0.2 % of such a call [MI355X, cfg 2,
 *    1024 steps: 1.423 against 1.426 G agent-steps/s] —, DEFAULT below (the 20 steps of the driver's bench command: 846
 *    against 1016 M agent-steps/s); 1: always STRICT; 0: never (but for captures).  A THIRD-PARTY BINDING should call
 *    riab_streamer_warmup once per (streamer, stream) at set-up and leave the option at 2 — or set 1 if it must never
 *    touch process-wide state, at the price above for short calls.
 * The per-kernel entry points above and the step-plan entry points conform unconditionally. */
int riab_streamer_warmup(RiabStreamer* h, riab_stream_t stream);
int riab_simulate(RiabStreamer* h, const RiabSimulate* run, riab_stream_t stream);
/* what riab_simulate does with RiabSimulate.watch before anything else (host memory only, no device): RIAB_OK when every
 * live array equals its snapshot, RIAB_ECHANGED otherwise, RIAB_EINVAL for a malformed list */
int riab_watch_compare(const RiabWatch* watch, int32_t n);
/* after a riab_simulate call with timed_pop >= 0 and after the caller has synchronised: the duration of the timed
 * population's kernel(s) in ms; < 0 if unavailable */
float riab_streamer_last_rate_ms(RiabStreamer* h);
/* which form the rate stage of the last riab_simulate call through `h` took */
enum { RIAB_FORM_NONE = 0, RIAB_FORM_ONE_KERNEL = 1, RIAB_FORM_CHUNKS = 2, RIAB_FORM_SERIAL = 3, RIAB_FORM_POPULATIONS = 4,
       RIAB_FORM_HEAD_AND_PIECES = 5 };
int riab_streamer_last_form(RiabStreamer* h);
/* what the form selection currently compares (which: 0 the trajectory step in ns, 1 the lead's store rate in MB/s, 2
 * whether they were measured (1) or are the built-in constants / configured values (0), 3 launches of the last call),
 * and how the trajectory kernel's stream was chosen: 4 the host time in ns of [one tiny launch on the caller's stream,
 * one on that stream, synchronise both] (best of six; -1: not screened yet), 5 how many candidate streams were rejected
 * before it (a stream that shares the caller's hardware queue makes that pair 30-50 us slower), 6 the same time with
 * both launches on the caller's stream; 7 calls whose rate stage the HOST launched more than 12 us after the trajectory
 * kernel's launch had returned (a descheduled thread, a kernel's first launch in the process): such a call can be counted
 * in ctrl[RIAB_CTRL_SERIALISED] without any queue being shared — readers subtract; 8 the last call ran in strict mode, 9
 * the streamer has its own second stream (riab_streamer_warmup), 10 the HIP streams the process-wide pool holds (bounded:
 * eight callers' streams per device have an entry, a ninth takes over the least recently used one) */
int64_t riab_streamer_info(RiabStreamer* h, int32_t which);

/* A/B switches of the library (comparisons and tests; the defaults are what production runs): process-wide, read on
 * every call they affect (plain loads of an int: the library never calls getenv).  Returns the previous value, or
 * RIAB_EINVAL for an unknown option / value.
 *   RIAB_OPT_TRAJ_KERNEL   which kernel multi-step riab_agent_step launches and riab_simulate's trajectory stage use:
 *                          0 (default) four specialised waves per 64 agents (csrc/riab_traj4_kernel.h); 1 the single-wave
 *                          kernel for every launch; 2 round 1's two-wave kernel (Philox launches of >= 32 steps)
 *   RIAB_OPT_FUSED_TASK    1 (default) a task plan's motion + task step is one launch; 0 two launches
 *   RIAB_OPT_BVC_BOX       1 (default) box fast path of the boundary-vector ray stage; 0 the general stage everywhere
 *   RIAB_OPT_NT_STORES     0 (default) only the one-kernel rate stage of riab_simulate writes its rows with nontemporal
 *                          stores; 1 the ungated PlaceCells / GridCells / HDC kernel too (A/B: measured slower)
 *   RIAB_OPT_PUB_SINGLE_ROWS  how many of a publishing trajectory launch's first rows leave one by one before blocks
 *                          of four (default 4; 0 .. 64)
 *   RIAB_OPT_POLL_SLEEP    the longest s_sleep between two polls of a waiting rate wave, in units of 64 cycles (default
 *                          48; 1 .. 127)
 *   RIAB_OPT_FUSED_STEP    1 (default) a plan that was given arrival words (riab_plan_set_fused) advances the agent and
 *                          its largest store-bound population in ONE launch per step; 2 the same with ordinary instead
 *                          of nontemporal stores; 0 the motion kernel and every population's kernel one after the other
 *   RIAB_OPT_STEP1_SPIN    log2 of the polls (~0.3 us each) a wait of the one-launch step makes before it gives up
 *                          (default 22: about a second; 0 .. 30.  Tests: 0 makes every wait that is not already
 *                          satisfied give up, which the host layer must recover from)
 *   RIAB_OPT_STEP1_RESIDENCY  1 (default) the one-launch step's grid follows the compute units the plan counts on
 *                          (riab_plan_set_compute_units) and a task plan's form is refused where it would not be
 *                          resident at once; 0 the grid of a whole, idle MI355X whatever the device (tests) */
enum { RIAB_OPT_TRAJ_KERNEL = 0, RIAB_OPT_FUSED_TASK = 1, RIAB_OPT_BVC_BOX = 2, RIAB_OPT_NT_STORES = 3,
       RIAB_OPT_PUB_SINGLE_ROWS = 4, RIAB_OPT_POLL_SLEEP = 5, RIAB_OPT_FUSED_STEP = 6, RIAB_OPT_STEP1_SPIN = 7,
       RIAB_OPT_STEP1_RESIDENCY = 8, RIAB_OPT_COUNT = 9 };
int riab_set_option(int32_t option, int32_t value);

/* Process-level host setting for latency-bound callers (one short simulate() per synchronisation, as in bench.py's
 * 20-step region): on != 0 makes the calling thread SPIN on completion signals in hipDeviceSynchronize /
 * hipStreamSynchronize (hipSetDeviceFlags(hipDeviceScheduleSpin)) instead of blocking in the driver after ~100 us of
 * active waiting; 0 restores the runtime's default.  Affects the current device of the calling thread. */
int riab_host_wait_spin(int32_t on);

/* sizeof of the ABI's structs as compiled into the library (which: 0 RiabEnv, 1 RiabMotion, 2 RiabRateIO,
 * 3 RiabPopulation, 4 RiabTask, 5 RiabFFInput, 7 RiabSimulate, 8 RiabWatch; 6 returns RIAB_TS_ROWS): bindings verify their mirrors at
 * load */
int64_t riab_abi_sizeof(int32_t which);

/* Streaming-store calibration kernel: writes `bytes` bytes (multiple of 16) of
 * a constant with the same 16-B/lane store pattern as the rate kernels.  Used
 * to calibrate the WRITE_SIZE counter and to measure the store roofline. */
int riab_fill(void* dst, int64_t bytes, float value, riab_stream_t stream);

int riab_abi_version(void);
const char* riab_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* RIAB_HIP_H */
