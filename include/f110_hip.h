/*
 * f110_hip.h -- C ABI of the MI355X-native batched F1TENTH step path.
 *
 * This is the drop-in boundary for the reference's hot path
 *   F110Env.step / reset          gym/f110_gym/envs/f110_env.py:261-347
 *   Simulator.step / reset        gym/f110_gym/envs/base_classes.py:546-623
 *   ScanSimulator2D.scan/set_map  gym/f110_gym/envs/laser_models.py:383-454
 * The reference has no FFI layer (it is Python + Numba); these are the entry
 * points a ctypes binding of that path needs (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success or a negative F110_E_* code and never
 *     throws; f110_last_error() gives the message of the calling thread's last
 *     failure.
 *   - plain pointers and sizes only.  "dev" pointers are device (HBM) addresses
 *     owned by the caller (e.g. torch tensors' data_ptr()); "host" pointers are
 *     read during the call and not retained.  Nothing is allocated in
 *     f110_step/f110_reset; kernels are enqueued on `stream` (a hipStream_t
 *     passed as void*, NULL = default stream) and the call does not synchronise.
 *   - a handle is bound to one device and is not thread-safe.  Calls that allocate or upload (f110_create, map
 *     installs, table uploads, f110_graph_create, f110_destroy ...) make the handle's device current for their own
 *     duration and RESTORE the caller's current device before returning.  Calls that launch on the caller's stream
 *     (f110_step, f110_reset, f110_graph_launch, the function-level entry points, f110_bitmap_render) do not switch:
 *     the stream belongs to the calling thread's current device, so that must be the handle's -- otherwise
 *     F110_E_INVALID.  Several handles (on one device or on several) may be driven from one process.
 *   - there is no f110_get_state / f110_set_state: the whole simulation state lives in the CALLER-owned
 *     buffers of the f110_buffers struct, bound once with f110_bind, so reading, checkpointing or overwriting the state is
 *     an ordinary access to the caller's own memory between steps (F110VecEnv.state_dict / load_state_dict).
 *   - tuning knobs read from the environment at first use (sweeps only; the defaults are the measured optimum):
 *     F110_WPC = 1|2|4|8 wavefronts per car in the scan kernel; F110_STAGES = "cars:log2waves,..." wave -> car
 *     stage list of a scan launch ("*" = the remaining cars), see launch_scan in csrc/f110_step.hip.
 *   - all arithmetic that decides an index, a collision or a lap toggle is
 *     IEEE fp64 in the reference's operation order (no FMA contraction).
 */
#ifndef F110_HIP_H
#define F110_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F110_OK 0
#define F110_E_INVALID (-1)  /* bad argument / state (ValueError in the reference) */
#define F110_E_HIP (-2)      /* HIP runtime failure */
#define F110_E_NOMAP (-3)    /* scan before set_map (laser_models.py:445-446) */
#define F110_E_INDEX (-4)    /* agent index out of range (base_classes.py:525-527) */
#define F110_E_UNBOUND (-5)  /* step/reset before f110_bind */

#define F110_MAX_CARS (1 << 26) /* num_envs * num_agents of one handle (32-bit wave / lane indices; offsets into the
                                  * per-car tensors are 64-bit): 67 M cars, 290 GB of fp32 scans alone at 1080 beams */
#define F110_MAX_AGENTS 32
#define F110_MAX_MAPS 4096  /* map slots of one handle (f110_set_map_slot_*, f110_assign_maps) */
#define F110_MAX_NOISE_SLOTS 64 /* noise slots (= distinct seeds) of one handle (f110_set_noise_generator, f110_assign_noise) */
#define F110_NOISE_INITIAL_ROWS 1024 /* rows per slot of a generated noise table when it is first allocated (it doubles on demand) */
#define F110_NUM_PARAMS 18
#define F110_RK4 1   /* Integrator.RK4   base_classes.py:40-42 */
#define F110_EULER 2 /* Integrator.Euler */

typedef struct f110_handle f110_handle;

/* Constructor arguments of F110Env (f110_env.py:100-157), Simulator
 * (base_classes.py:459) and ScanSimulator2D (laser_models.py:360), plus the
 * batch extension (num_envs, autoreset, device). */
typedef struct {
    int32_t num_envs;    /* B: independent envs on this device */
    int32_t num_agents;  /* A: cars per env (1..F110_MAX_AGENTS) */
    int32_t num_beams;   /* 1080 */
    int32_t theta_dis;   /* 2000 */
    int32_t integrator;  /* F110_RK4 | F110_EULER */
    int32_t ego_idx;
    int32_t device;      /* HIP device ordinal */
    int32_t autoreset;   /* 1: an env that reports done is reset to its spawn pose by the next f110_step */
    double fov;          /* 2*pi */
    double eps;          /* 1e-4 */
    double max_range;    /* 30.0 */
    double timestep;     /* 0.01 */
    double ttc_thresh;   /* 0.005 (base_classes.py:113) */
    /* mu C_Sf C_Sr lf lr h m I s_min s_max sv_min sv_max v_switch a_max v_min v_max width length */
    double params[F110_NUM_PARAMS];
} f110_config;

/* Caller-owned device buffers the step reads and writes (N = B*A cars).
 * Simulation state (carried from step to step): */
typedef struct {
    double *state;          /* [N,7]  x y steer v yaw yaw_rate slip   base_classes.py:95-96 */
    double *steer_buf;      /* [N,2]  steering delay FIFO, [0] newest  :269-276 */
    int32_t *steer_cnt;     /* [N]    entries in the FIFO (0..2) */
    int32_t *noise_step;    /* [N]    scans drawn since the car's reset (row of the noise table) */
    double *spawn;          /* [N,3]  pose used by reset / autoreset */
    double *start_rot;      /* [B,4]  f110_env.py:329 */
    uint8_t *near_start;    /* [N]    f110_env.py:182 */
    int32_t *toggles;       /* [N]    f110_env.py:183 toggle_list */
    double *current_time;   /* [B]    f110_env.py:177 */
    uint8_t *pending_reset; /* [B]    1: next f110_step performs reset(spawn) + zero-action step for this env */
    /* observations (overwritten every step): */
    float *scans;           /* [N,num_beams] fp32 lidar ranges (noise, opponents applied) */
    double *scans_f64;      /* [N,num_beams] same in fp64, or NULL to skip (parity tests, single-env facade) */
    double *pose_snap;      /* [N,3]  poses after integration, before iTTC zeroing (base_classes.py:567) */
    uint8_t *collisions;    /* [N]    obs['collisions'] (GJK or iTTC)  :543,:581-582 */
    int32_t *collision_idx; /* [N]    Simulator.collision_idx, -1 = none */
    uint8_t *in_collision;  /* [N]    RaceCar.in_collision (iTTC only) */
    int32_t *lap_counts;    /* [N]    f110_env.py:238 */
    double *lap_times;      /* [N]    f110_env.py:240 */
    uint8_t *done;          /* [B]    f110_env.py:242 (0/1, so the buffer may be a bool tensor) */
    uint8_t *checkpoint_done; /* [N]  info['checkpoint_done'] = toggles >= 4 (f110_env.py:244), or NULL */
    uint32_t *lookups;      /* [N]    distance-table reads per car, ACCUMULATED over steps until the caller
                                      zeroes it (instrumentation for the byte model), or NULL */
} f110_buffers;

int f110_create(const f110_config *cfg, f110_handle **out);
void f110_destroy(f110_handle *h);
const char *f110_last_error(void);

/* Simulator.update_params (base_classes.py:507-527): agent_idx < 0 updates every agent,
 * otherwise agent `agent_idx` of every env (F110_E_INDEX if >= num_agents).  As in the
 * reference this changes the cars' dynamics and the size they attribute to opponents
 * (base_classes.py:221), not the Simulator's own copy used by the GJK check (:542) nor
 * the beam tables fixed at construction (:116-156). */
int f110_update_params(f110_handle *h, const double *params18_host, int32_t agent_idx);

/* Per-env constructor arguments.  One handle stands in for num_envs F110Env instances; the reference constructs each
 * with its own `params` (f110_env.py:125-128) and `seed` (:102-105).  Like maps (slots + f110_assign_maps), both exist
 * as SLOTS plus an env -> slot table:
 *   params slot = the `params` dict of one reference env: Simulator.params (the GJK quads, base_classes.py:542) and the
 *     RaceCar.params of every agent (:84,169).  f110_set_params_slots replaces the whole slot table ([n_slots,18] host,
 *     1 <= n_slots <= num_envs: "every env its own vehicle" is n_slots = num_envs with the identity assignment);
 *     f110_set_params_slot changes one slot -- agent_idx < 0: as at construction (Simulator copy + every agent), else only
 *     RaceCar.params of that agent (Simulator.update_params on that env, :507-527); a slot beyond the current count extends
 *     the table with copies of slot 0.  f110_update_params (above) keeps applying to every slot.  f110_assign_params: host
 *     int32 [num_envs], NULL = all envs on slot 0.  The beam tables (scan angles, cosines, side distances) stay per handle:
 *     in the reference they are class-level statics fixed by the FIRST RaceCar a process constructs (base_classes.py:116-156).
 *   noise slot = one seed: see f110_set_noise_generator / f110_set_noise_slot / f110_assign_noise below. */
int f110_set_params_slots(f110_handle *h, const double *params_host, int32_t n_slots);
int f110_set_params_slot(f110_handle *h, int32_t slot, const double *params18_host, int32_t agent_idx);
int f110_assign_params(f110_handle *h, const int32_t *slot_of_env_host);

/* Optional: replace the library's libm-computed tables by the caller's
 * (numpy-computed, as in the reference).  sines/cosines: [theta_dis]
 * (laser_models.py:379-381); scan_angles/beam_cosines/side_distances:
 * [num_beams] (base_classes.py:123-156).  Any pointer may be NULL = keep. */
int f110_set_tables(f110_handle *h, const double *sines_host, const double *cosines_host,
                    const double *scan_angles_host, const double *beam_cosines_host,
                    const double *side_distances_host);

/* ScanSimulator2D.set_map (laser_models.py:383-427) after image decoding.
 * free_mask: [H*W] row-major, row 0 = bottom of the image (already flipped,
 * :399), nonzero = free (>128, :403-404).  The library runs an exact Euclidean
 * distance transform (replaces scipy.ndimage.distance_transform_edt, :52). */
int f110_set_map_occupancy(f110_handle *h, const uint8_t *free_mask_host, int32_t height,
                           int32_t width, double resolution, double orig_x, double orig_y,
                           double orig_c, double orig_s);
/* Same with the mask already on the device (e.g. drawn by a track generator): the whole pipeline -- EDT,
 * rank coding, LUT, fp64 table -- runs on the GPU; only two scalars and the 8 KiB LDS image of the LUT visit
 * the host.  An all-free mask (no occupied cell) is the caller's responsibility here. */
int f110_set_map_occupancy_dev(f110_handle *h, const uint8_t *free_mask_dev, int32_t height, int32_t width,
                               double resolution, double orig_x, double orig_y, double orig_c, double orig_s);
/* Walls of a generated track (replaces the drawing half of unittest/random_trackgen.py:161-218): mask_dev
 * [H*W] gets 0 where the distance from the pixel centre (x0 + (ix+.5)*pixel, y0 + (iy+.5)*pixel) to the
 * polyline pts_dev [n_pts,2] (closed: last point joins the first) is within half_stroke of `offset`, else 1.
 * Feed the result to f110_set_map_occupancy_dev. */
int f110_track_mask(const double *pts_dev, int32_t n_pts, int32_t closed, int32_t height, int32_t width, double x0,
                    double y0, double pixel, double offset, double half_stroke, uint8_t *mask_dev, void *stream);
/* Map slots: one handle stands in for many F110Env instances, each of which may have its own map
 * (f110_env.py:100-157 takes `map` per env).  Slot 0 is the map of the calls above; f110_set_map_slot_* fill
 * slots 0..F110_MAX_MAPS-1 the same way, and f110_assign_maps gives every env its slot (host int32 [num_envs],
 * NULL = all envs on slot 0; F110_E_INDEX for a slot that holds no map).  Any assignment is valid.  The scan keeps its
 * full occupancy when the 2 consecutive cars of every scan workgroup share a map, i.e. maps assigned to blocks of envs
 * with an even car count; otherwise (a map per env) it runs one wave per workgroup, each staging its own map's table:
 * same bits, ~24 % fewer env-steps/s at 65 536 cars (profiles/r05_map_per_env.txt).  Maps of different kinds ("resolution is a power of two",
 * "origin unrotated") may be mixed: the shard is then scanned block by block, each run of envs with the instantiation
 * its own maps allow (one more launch per change of kind along the env index). */
int f110_set_map_slot_occupancy(f110_handle *h, int32_t slot, const uint8_t *free_mask_host, int32_t height, int32_t width,
                                double resolution, double orig_x, double orig_y, double orig_c, double orig_s);
int f110_set_map_slot_occupancy_dev(f110_handle *h, int32_t slot, const uint8_t *free_mask_dev, int32_t height,
                                    int32_t width, double resolution, double orig_x, double orig_y, double orig_c,
                                    double orig_s);
int f110_assign_maps(f110_handle *h, const int32_t *map_of_env_host);
int f110_get_map_slot_dt(f110_handle *h, int32_t slot, double *dt_host_out);
/* Same, from a precomputed distance table dt = resolution*edt(img) (host, [H*W] fp64). */
int f110_set_map_dt(f110_handle *h, const double *dt_host, int32_t height, int32_t width,
                    double resolution, double orig_x, double orig_y, double orig_c, double orig_s);
/* Copies the fp64 distance table the handle holds to the host (tests). */
int f110_get_map_dt(f110_handle *h, double *dt_host_out);

/* Exact squared Euclidean distance transform on the host (cells to nearest
 * zero cell), the integer kernel behind f110_set_map_occupancy. */
int f110_edt_squared(const uint8_t *free_mask_host, int32_t height, int32_t width, uint32_t *d2_out_host);
/* The same transform on the GPU (dev pointers; synchronises `stream` before returning): a column pass and a
 * row pass with the row's g^2 in LDS, exact like the host version.  height, width <= 32768. */
int f110_edt_squared_dev(const uint8_t *free_mask_dev, int32_t height, int32_t width, uint32_t *d2_out_dev, void *stream);

/* Lidar noise.  Reference: every scan adds `rng.normal(0, std, num_beams)` drawn from the car's own
 * np.random.default_rng(seed), re-created at every reset (laser_models.py:450-452, base_classes.py:117,202); all cars of an
 * env share the seed, so a car's noise row is a function of (seed, scans since its reset).  The handle keeps the rows of
 * each seed ONCE, in a noise slot, indexed by every car's own counter (f110_buffers.noise_step); f110_assign_noise gives
 * every env its slot (host int32 [num_envs], NULL = all on slot 0).  A slot is filled either way:
 *   f110_set_noise_generator -- the rows are PRODUCED ON THE DEVICE by a bit-level restatement of NumPy's PCG64 +
 *     ziggurat `normal` (csrc/f110_noise.h).  pcg64_state_inc = {state_lo, state_hi, inc_lo, inc_hi} of
 *     np.random.PCG64(seed).state['state'] (NumPy's SeedSequence hashing stays with NumPy; a C caller may pass any
 *     128-bit state and odd increment).  Setting a generator restarts every generated slot at row 0.
 *   f110_set_noise_slot -- rows [T,num_beams] fp64 from the host (any table; tests, non-NumPy streams).
 *     f110_set_noise_table(h, tbl, T) is slot 0; T = 0 / NULL switches noise off for the whole handle.
 * Rows: f110_noise_ensure(h, rows, stream) makes rows 0 .. rows-1 (above the floor) readable for work enqueued on
 * `stream` afterwards -- call it before a step with rows > the largest noise_step any car can have in that step;
 * generated slots produce what is missing (a one-wavefront kernel per slot; the table doubles when it must: cold path,
 * synchronises), host tables that are too short give F110_E_INVALID.  f110_noise_prefetch(h, rows) starts producing
 * ahead of time on the handle's own stream, beside the caller's work; a later f110_noise_ensure only waits for it.
 * Floor: when no car will read rows below `lo` in any step enqueued FROM NOW ON (autoreset off and, by the host's own step
 * count, every car past them), f110_noise_set_floor(h, lo, stream) lets the table recycle them -- it is a ring, so a run of any
 * length holds a window of rows in constant memory.  Steps already enqueued on `stream` may still read those rows: the call
 * records an event there, and the next prefetch (which writes the recycled places from the handle's own stream) waits for it;
 * so does a generator launch that f110_noise_ensure put on the caller's stream (both work on the same generator states).
 * Lowering the floor again (a masked reset sends cars back to row 0 while others run on) re-produces the dropped rows in
 * `stream` from the MARKS the generators leave every 64 rows -- one wavefront per slot and 64 rows, all at once; the generators
 * stay where they are (round 4 rewound them to the seeds: ~15 us per row of the whole run).  The ring then spans floor .. rows
 * produced, i.e. the ages of the youngest and the oldest car: rows x num_beams x 8 B per slot (20 000 steps: 173 MB per seed).
 * A car whose row lies outside [floor, rows produced) sets F110_DEVERR_NOISE_WINDOW in the device error word instead of
 * reading silently.
 * Launch epoch: the scan takes the table's base and size by value (a pointer chase per wave costs 0.9 % of the launch), so
 * a RE-ALLOCATION -- the ring too small for the rows between the floor and the fastest car: it doubles -- moves the launch
 * epoch like every other table change; the window of rows present (floor, rows produced) moves without it, behind a
 * device-resident descriptor.  A run whose floor follows its cars never re-allocates: one captured hipGraph serves it
 * for any length (tests/test_gpu_noise.py: 20 000 replays). */
int f110_set_noise_table(f110_handle *h, const double *table_host, int64_t T);
int f110_set_noise_slot(f110_handle *h, int32_t slot, const double *table_host, int64_t T);
int f110_set_noise_generator(f110_handle *h, int32_t slot, const uint64_t *pcg64_state_inc, double std_dev);
int f110_assign_noise(f110_handle *h, const int32_t *slot_of_env_host);
/* Every env its own seed, any number of them (F110_MAX_NOISE_SLOTS does not apply): pcg64_state_inc_host = [num_envs][4]
 * {state_lo, state_hi, inc_lo, inc_hi} of np.random.PCG64(seed_e).state['state'].  The env's generator state lives on the
 * device; the row its scan adds is produced by the step itself, in front of the scan (one wavefront per env and step), into
 * ONE row per env -- no table to keep ahead of the cars, no floor, no growth: f110_noise_ensure / _prefetch / _set_floor
 * are no-ops in this mode.  A reset restarts the env's stream at its seed; a row counter that does not continue the
 * generator's (a loaded checkpoint) makes the generator run forward from the seed without storing.  Costs one more kernel
 * per step (~1 500 wave-instructions per env).  Setting a table or a slot generator leaves the mode.
 * Reference: f110_env.py:102-105 (`seed` of every env), base_classes.py:117,202, laser_models.py:450-452. */
int f110_set_noise_per_env(f110_handle *h, const uint64_t *pcg64_state_inc_host, double std_dev);
int f110_noise_ensure(f110_handle *h, int64_t rows, void *stream);
int f110_noise_prefetch(f110_handle *h, int64_t rows);
int f110_noise_set_floor(f110_handle *h, int64_t lo, void *stream);
/* floor, rows readable by the kernels (a prefetch still in flight not counted), rows per slot of the ring, slots, bytes held
 * (host-side bookkeeping, no synchronisation; any pointer may be NULL) */
int f110_noise_info(f110_handle *h, int64_t *lo, int64_t *hi, int64_t *cap, int32_t *slots, int64_t *bytes);
/* Tests / diagnostics: the noise values of rows row0 .. row0+n_rows-1 of a slot, [n_rows,num_beams] fp64 to the host
 * (synchronises; F110_E_INDEX when a row is not in the table). */
int f110_noise_read(f110_handle *h, int32_t slot, int64_t row0, int64_t n_rows, double *out_host);

/* Device error word: conditions a kernel can only detect while it runs are OR-ed into one word per handle instead of
 * being silent.  Synchronises the device, returns the word and clears it.  F110_DEVERR_NOISE_WINDOW: a car's noise row
 * was not in the table (f110_noise_ensure not called, or the floor above a live car).  F110_DEVERR_BOUNDS: only in the
 * bounds-checked debug build of the library (-DF110_BOUNDS, tools/build_variant.sh): an index into a device table was out
 * of range; bits 8.. name the table (csrc/f110_kernels.h BOUNDS_*). */
#define F110_DEVERR_NOISE_WINDOW 0x1u
#define F110_DEVERR_BOUNDS 0x2u
int f110_device_errors(f110_handle *h, uint32_t *flags_out);

int f110_bind(f110_handle *h, const f110_buffers *bufs);

/* F110Env.reset(poses) (f110_env.py:304-347) for the envs selected by mask
 * (dev [B] uint8, NULL = all): stores poses as spawn, resets those envs and
 * runs their zero-action step; other envs are untouched.  poses: dev [B,A,3]. */
int f110_reset(f110_handle *h, const double *poses_dev, const uint8_t *mask_dev, void *stream);

/* F110Env.step(action) for all B envs (f110_env.py:261-302).  actions: dev
 * [B,A,2] fp64, column 0 = steer, column 1 = speed.  Envs with pending_reset
 * ignore their action and perform reset(spawn) + zero-action step instead. */
int f110_step(f110_handle *h, const double *actions_dev, void *stream);

/* One env's observation gathered into ONE fp64 row on the device (the single-env Gym facade copies it to the host with
 * one transfer instead of one per field): out_dev[f110_pack_env_size(h)] =
 *   [A*7 state | A collisions | A lap_times | A lap_counts | A toggles | current_time | done | A*num_beams scans]
 * (scans from scans_f64 when bound, else the fp32 scans widened).  Enqueued on `stream`, no synchronisation. */
int64_t f110_pack_env_size(f110_handle *h);
int f110_pack_env(f110_handle *h, int32_t env, double *out_dev, void *stream);

/* Tuning / test hook: how a scan launch maps wavefronts to cars, as "cars:lg,cars:lg,..." in launch order with one
 * "*" for the remaining cars: a car of a stage gets 2^lg wavefronts (lg = 0..3; several short-lived waves per car pay
 * for small batches and at the end of a launch).  NULL or "" restores the built-in choice (or the F110_STAGES
 * environment variable).  A malformed list (syntax, lg > 3, two "*", more than 6 stages, more cars than the handle
 * has) is refused with F110_E_INVALID and changes nothing.  Results do not depend on it. */
int f110_set_scan_stages(f110_handle *h, const char *spec);

/* hipGraph support.  f110_step only enqueues kernels (no allocation, no synchronisation), so it can be captured
 * into a HIP graph and replayed.  A capture freezes the kernel selection and the by-value launch arguments; the
 * calls that change them -- f110_bind, f110_set_tables, every map install, f110_assign_maps / _params / _noise, a
 * re-allocation of the noise table -- bump the handle's launch epoch.  A graph captured at epoch e is valid while
 * f110_launch_epoch still reports e; after that it must be re-captured (F110VecEnv.step_graph does so itself). */
/* Launch order of the step's scan (performance only; no reference counterpart): order_dev = dev int32 [num_envs * num_agents], a
 * PERMUTATION of the car indices (the caller's responsibility: a wrong array scans some cars twice and others not at all), or NULL
 * = car order.  The wave that would march car i marches car order[i]; every result is stored under the car's own index, so the
 * outputs do not depend on it.  Purpose: cars that stand on the same noise row launched side by side share that row in the L1 / L2 --
 * a batch whose envs were reset at different times otherwise streams one 8.6 KB row per env from HBM (10 % of the step at 65 536
 * envs).  The array stays owned by the caller and may be re-written in stream order between steps (red_gym_amd.Engine re-sorts it
 * by the envs' row counters every 64 steps); changing the POINTER moves the launch epoch.  Ignored while envs are on different maps. */
int f110_set_scan_order(f110_handle *h, const int32_t *order_dev);
int f110_launch_epoch(f110_handle *h, int64_t *epoch);

/* The step as a HIP graph the library builds itself, for callers without a capturing framework and as the reference
 * point for framework captures: F110_GRAPH_NODES = one kernel node per kernel of f110_step(actions_dev), chained
 * (hipGraphAddKernelNode); F110_GRAPH_CAPTURE = the same launches captured on a private non-blocking stream.
 * f110_graph_launch enqueues one step on `stream` (the actions are read from actions_dev at execution time) and
 * refuses a graph whose handle's launch epoch has moved (F110_E_INVALID: create it again).  f110_graph_info reports
 * the node count and, with a path, writes the graph in dot form (hipGraphDebugDotPrint). */
#define F110_GRAPH_NODES 0
#define F110_GRAPH_CAPTURE 1
typedef struct f110_graph f110_graph;
int f110_graph_create(f110_handle *h, const double *actions_dev, int32_t how, f110_graph **out);
int f110_graph_launch(f110_graph *g, void *stream);
int f110_graph_info(f110_graph *g, int32_t *nodes, const char *dot_path);
void f110_graph_destroy(f110_graph *g);

/* Measurement aid (bench.py): between begin and end every f110_step attaches a start / stop
 * hipEvent pair to its scan_kernel dispatch on the step's stream (up to max_launches
 * steps).  f110_profile_end synchronises on the last event and returns the summed
 * kernel time in milliseconds and the number of launches measured. */
int f110_profile_begin(f110_handle *h, int32_t max_launches);
/* Sampling: events ride on every `every`-th step only (default 1 = all).  A dispatch that carries events costs the
 * stream about 10 us of idle time around it (measured, profiles/r03_event_cost.txt), which a 4 096-env step notices. */
int f110_profile_every(f110_handle *h, int32_t every);
/* While the aid is active (between begin and end) f110_buffers.lookups is only fed by the steps that carry an event
 * pair: the counters then hold the table reads of exactly the measured launches. */
int f110_profile_end(f110_handle *h, double *scan_ms_total, int32_t *launches);

/* Batched pure-pursuit planner, the caller on the other side of F110Env.step
 * (examples/waypoint_follow.py:15-217: nearest point on the raceline, first intersection
 * of the lookahead circle with the polyline incl. wrap-around, actuation).  waypoints:
 * dev [M,3] (x, y, speed); state: dev [n,7] (f110_buffers.state); writes actions dev [n,2]
 * = (steer, vgain*speed), ready to be passed to f110_step.  The planner keeps no state: h may be NULL (the
 * kernel is then enqueued on `stream` of the calling thread's current device). */
/* Optional: PREPARES one raceline (dev [M,3], M <= 65 535) for f110_pure_pursuit -- a grid of `cell` metres (0: 0.25) reaching
 * `margin` metres (0: 3) around the raceline whose cells list the segments that can be the nearest one for any pose in the cell
 * (conservative: every segment within 2 half-diagonals of the cell centre's nearest).  f110_pure_pursuit(h, the same pointer,
 * the same M, ...) then plans with ONE LANE per car over that list instead of one wavefront per car over 64-segment blocks
 * (65 536 cars on the 783-point example raceline: see profiles/r05_planner.txt); poses outside the grid, NaN poses and
 * cells with more than 30 candidates take every segment -- the results are the same in every case (tests: `==` both
 * kernels and oracle/planner.py).  A cold path (copies the raceline to the host, synchronises); call it again when the raceline's
 * VALUES change.  Reference: examples/waypoint_follow.py:15-47 (nearest point), :183-217 (plan). */
int f110_pure_pursuit_prepare(f110_handle *h, const double *waypoints, int32_t M, double cell, double margin, void *stream);
int f110_pure_pursuit(f110_handle *h, const double *waypoints, int32_t M, double lookahead, double vgain,
                      double wheelbase, double max_reacquire, const double *state, int32_t n,
                      double *actions, void *stream);

/* The same planner for MANY racelines in one launch, of any length (the reference builds one planner per env from any
 * CSV, examples/waypoint_follow.py:146-162): waypoints dev [total,3] = the K racelines back to back; offsets [K+1] =
 * first row of each (offsets[0] = 0, offsets[K] = total) given both as a dev and as a host array; track_of_car dev
 * [n] int32 = raceline of every car (NULL: all on raceline 0).  workspace: dev doubles,
 * f110_pure_pursuit_workspace(total, K) of them, holding the bounding boxes of the racelines' 64-segment blocks;
 * boxes_valid != 0 says the workspace still holds them from an earlier call with the same racelines (skips the small
 * kernel that fills it).  Racelines are read from global memory (L1 / L2), so there is no length limit.  A raceline
 * with two equal consecutive waypoints gives (steer 0, speed 4.0) for every car on it, as the reference does (NaN
 * nearest distance).  f110_pure_pursuit itself stages its raceline in LDS while it fits (gfx950: about 6 400 points)
 * and otherwise runs this form without a workspace (every block evaluated). */
int64_t f110_pure_pursuit_workspace(int32_t total_points, int32_t K);
int f110_pure_pursuit_tracks(f110_handle *h, const double *waypoints, const int32_t *offsets_dev,
                             const int32_t *offsets_host, int32_t K, const int32_t *track_of_car, double lookahead,
                             double vgain, double wheelbase, double max_reacquire, const double *state, int32_t n,
                             double *actions, double *workspace, int32_t boxes_valid, void *stream);

/* ---- function-level entry points (parity tests; all pointers dev) ---- */
/* ScanSimulator2D.scan(pose, None): n poses [n,3] -> [n,num_beams] (noise off).
 * scans_f32 / lookups may be NULL; lookups [n] is overwritten-by-accumulation like
 * f110_buffers.lookups (zero it first). */
int f110_scan(f110_handle *h, const double *poses, int32_t n, double *scans_f64, float *scans_f32,
              uint32_t *lookups, void *stream);
/* RaceCar.update_pose without the scan (base_classes.py:254-402), n cars in place. */
int f110_update_pose(f110_handle *h, double *state, double *steer_buf, int32_t *steer_cnt,
                     const double *actions, int32_t n, void *stream);
/* vehicle_dynamics_st (dynamic_models.py:124-176; kinematic == 1: vehicle_dynamics_ks
 * :91-121 on the first 5 states): right-hand sides f [n,7] for states x [n,7] and inputs
 * u [n,2] = (steering velocity, acceleration), with the handle's agent-0 parameters. */
int f110_vehicle_dynamics(f110_handle *h, const double *x, const double *u, int32_t n, int32_t kinematic,
                          double *f, void *stream);
/* get_vertices (collision_models.py:238-260): [n,3] -> [n,4,2] */
int f110_get_vertices(f110_handle *h, const double *poses, int32_t n, double *verts, void *stream);
/* collision (GJK, collision_models.py:114-182) on n quad pairs -> hit[n] */
int f110_gjk_pairs(f110_handle *h, const double *verts_a, const double *verts_b, int32_t n,
                   uint8_t *hit, void *stream);
/* collision_multiple (collision_models.py:185-212): n groups of A quads [n,A,4,2] */
int f110_collision_multiple(f110_handle *h, const double *verts, int32_t n, int32_t A,
                            uint8_t *collisions, int32_t *collision_idx, void *stream);
/* check_ttc_jit (laser_models.py:189-217): scans [n,num_beams], vel [n] -> hit[n] */
int f110_check_ttc(f110_handle *h, const double *scans, const double *vel, int32_t n, uint8_t *hit,
                   void *stream);
/* ray_cast (laser_models.py:319-346): scans [n,num_beams] modified in place by one
 * opponent quad each; span [n,2] = get_blocked_view_indices (may be NULL). */
int f110_ray_cast(f110_handle *h, const double *ego_poses, const double *opp_verts, int32_t n,
                  double *scans, int32_t *span, void *stream);

/* F110Env._check_done (f110_env.py:202-244) on n envs of num_agents cars, stateless (h may be NULL):
 * poses [n,A,3] (x, y, theta) and start_poses [n,A,3] (the poses given to reset: start_xs / start_ys, :321-323),
 * start_rot [n,4] the EGO's 2x2 start rotation, row-major (:329), current_time [n] (already advanced by the
 * step, :293), collisions [n,A] (only the ego's entry is read, :242).  In/out: near_start [n,A] (0/1, True after
 * reset), toggles [n,A], lap_times [n,A] (frozen once a car has 4 toggles).  Out: lap_counts [n,A] = toggles // 2,
 * done [n] = collisions[ego] or all(toggles >= 4), checkpoint_done [n,A] = toggles >= 4 (may be NULL).
 * f110_step runs the same device function inside its env kernel. */
int f110_check_done(f110_handle *h, const double *poses, const double *start_poses, const double *start_rot,
                    const double *current_time, const uint8_t *collisions, int32_t n, int32_t num_agents,
                    int32_t ego_idx, uint8_t *near_start, int32_t *toggles, int32_t *lap_counts, double *lap_times,
                    uint8_t *done, uint8_t *checkpoint_done, void *stream);

/* ---- scan -> bird's-eye bitmap (the first consumer of the step's scans) ----
 * Replaces weap_util/weap_util/lidar.py:105-154 `lidar_to_bitmap` (same body in src/SAL.py:274-395
 * and src/bitmap.py:4-140), which draws one scan with OpenCV 4.11 (fillPoly / polylines / line /
 * rectangle, 8-bit, LINE_8).  A renderer holds the scan-independent tables the reference rebuilds
 * per call (lidar.py:63-72): indices = np.linspace(0, num_beams-1, T, dtype=int) and cos / sin of
 * angles = starting_angle + dir*fov*np.linspace(0, 1, T) -- computed by the caller with numpy so that
 * they are the reference's own values (host pointers, copied). */
#define F110_BITMAP_FILL 0
#define F110_BITMAP_POLYGON 1
#define F110_BITMAP_RAYS 2
typedef struct f110_bitmap f110_bitmap;
typedef struct {
    int32_t device;
    int32_t num_beams;          /* len(scan) */
    int32_t target_beam_count;  /* T, 0 < T < num_beams, T <= 2048 */
    int32_t rows, cols;         /* output_image_dims */
    int32_t channels;           /* 1, 3 or 4 (alpha = 255) */
    int32_t draw_mode;          /* F110_BITMAP_* */
    int32_t bg_value, draw_value; /* grey levels (0/255 white-on-black etc., lidar.py:61; 0/180 in src/bitmap.py:60) */
    int32_t draw_center;
    double scaling_factor;      /* pixels per metre (min(dims)/max_scan_radius when that is given) */
} f110_bitmap_config;
int f110_bitmap_create(const f110_bitmap_config *cfg, const int32_t *indices, const double *cosines,
                       const double *sines, f110_bitmap **out);
void f110_bitmap_destroy(f110_bitmap *b);
/* n scans (dev, f32 or f64, `stride` elements apart) -> out dev uint8 [n, rows, cols(, channels)].
 * Enqueued on `stream`; no allocation, no synchronisation. */
int f110_bitmap_render(f110_bitmap *b, const void *scans, int32_t scans_f64, int64_t n, int64_t stride,
                       uint8_t *out, void *stream);
/* Function-level view of the renderer's first stage (parity tests): the integer points the reference computes
 * at lidar.py:63-73 and hands to cv2.fillPoly / polylines / line -- points dev int32 [n, T, 2] = (x, y) of
 * np.rint(center + scaling_factor * scan[indices] * {cos, sin}(angles)).astype(int), center = (rows//2, cols//2). */
int f110_bitmap_points(f110_bitmap *b, const void *scans, int32_t scans_f64, int64_t n, int64_t stride,
                       int32_t *points, void *stream);
/* Point-occupancy grid of f1tenth_gym/examples/lidar.py:212-244 (the routine that wrote the
 * reference's lidar_datasets): out dev uint8 [n, grid, grid] of 0/1.  cosines / sines: dev [num_beams],
 * of np.linspace(-135, 135, num_beams) * pi / 180. */
int f110_scan_occupancy(const void *scans, int32_t scans_f64, int64_t n, int64_t stride, int32_t num_beams,
                        const double *cosines, const double *sines, double max_range, double lo, double hi,
                        int32_t grid, uint8_t *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* F110_HIP_H */
