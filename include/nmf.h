/*
 * nmf.h — C ABI of the MI355X-native batched NeuroMechFly stepping engine (libnmf_hip.so).
 *
 * The reference has no FFI for this path: its "operator API" is the Python class
 * flygym.warp.GPUSimulation (reference src/flygym/warp/simulation.py:28-453), which forwards
 * to mujoco_warp.  Each entry point below replaces one group of those calls; the Python class
 * flygym_amd.HIPSimulation binds them with ctypes (see INTEGRATION.md for the stub a flygym
 * maintainer would add).  No torch / HIP types appear in the signatures: device buffers are
 * plain pointers, streams are passed as void* (hipStream_t), sizes are ints.
 *
 * All device arrays are float32, row-major, world-major: field[n_worlds][width].
 * All calls are stream-ordered device work without host synchronisation (hipGraph-capturable)
 * unless stated otherwise.  Return value: 0 on success, negative on error (see nmf_last_error).
 */
#ifndef NMF_H_
#define NMF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nmf_model nmf_model;
typedef struct nmf_batch nmf_batch;
typedef struct nmf_eye_plan nmf_eye_plan;

/* Per-world fields addressable through nmf_field_ptr / nmf_gather_* */
enum nmf_field {
  NMF_QPOS = 0,         /* [nq]      generalized positions (free joint 7 + hinges)            */
  NMF_QVEL = 1,         /* [nv]                                                                */
  NMF_CTRL = 2,         /* [nu]      actuator controls                                         */
  NMF_QACC_WARMSTART = 3, /* [nv]                                                              */
  NMF_SEG_XPOS = 4,     /* [nseg*3]  world positions of the named body segments               */
  NMF_SEG_XQUAT = 5,    /* [nseg*4]  world orientations (w,x,y,z)                              */
  NMF_SITE_XPOS = 6,    /* [nsite*3]                                                           */
  NMF_ACTUATOR_FORCE = 7, /* [nu]                                                              */
  NMF_SENSORDATA = 8,   /* [96]      6 legs x (found, force3, torque3, pos3, normal3, tangent3)*/
  NMF_TIME = 9,         /* [1]                                                                 */
  NMF_STATS = 10,       /* [8]       of the launch's last step, as floats: ncon, solver iterations, overflow flag, nefc, solve-report
                                    bits (bit k = column 4 + k of NMF_STATS_SUM), most pivots of an elimination, KKT residual of
                                    the last elimination's target (0 = exact end; contact-space solve only), 0                */
  NMF_QACC = 11,        /* [nv]                                                                */
  NMF_COST = 12,        /* [1]       shader cycles the world took in the last stepping launch (load metric) */
  NMF_STATS_SUM = 13,   /* [16]      since the last reset, uint32_t COUNTERS (read them through a uint32_t / int32_t view of the
                                    pointer nmf_field_ptr returns; means over any window = differences of two reads; exact up to
                                    4.29e9, i.e. ~7e8 steps of one world at 6 contacts per step between two resets):
                                    0 physics steps, 1 sum of ncon, 2 sum of solver iterations, 3 steps with contact overflow;
                                    how the steps' constraint solves ended — 4 solved in contact space, of which 5 exactly (the
                                    elimination's target satisfies its own active set: KKT), 6 by the tie rule, 7 after a fourth
                                    line search without measurable descent, 8 by the cost tests, 9 at the iteration limit;
                                    10 solved by the primal Newton loop, of which 11 after a contact-space solve whose end failed
                                    the residual test; 12 steps with an elimination of more than 47 pivots; 13 steps with contacts
                                    that could not take the CPU flavour's noslip pass; 14 steps without contact; 15 unused    */
  NMF_CONTACT_GEOM = 14, /* [48]     contact list of the launch's last step: index (into the world's contact-geom list,
                                    reference compose/world.py:300-309 pair order sorted by body) of the geom of contact
                                    c, as a float; -1 beyond ncon                                                    */
  NMF_ACT = 15,         /* [nu]      activation state of the stateful actuators (intvelocity, cylinder, muscle: MuJoCo's act, here one
                                    slot per actuator — 0 for the stateless ones); reset to 0; writable between launches       */
  NMF_FIELD_COUNT = 16
};

/* Text of the last error raised on the calling thread ("" if none). */
const char* nmf_last_error(void);

/* Parse a compiled-model blob (flygym_amd.compiler.model.CompiledModel.to_blob()).
 * Replaces: world.compile() + mjw.put_model  (reference simulation.py:37, warp/simulation.py:417). */
nmf_model* nmf_model_create(const void* blob, size_t nbytes);
void nmf_model_destroy(nmf_model* model);

/* out[0..9] = nq, nv, nu, nbody, nseg, ngeom, nsite, max_contacts, nsensordata, is_star */
int nmf_model_dims(const nmf_model* model, int32_t out[10]);

/* The most contacts the model's contact set can make in one step (capsules: two end spheres, hulls: up to four vertices;
 * over a terrain with side faces one face contact more per probe).  The engine keeps at most 48 per world (out[7] of
 * nmf_model_dims).  Negative on error. */
int nmf_model_contact_bound(const nmf_model* model);

/* Allocate the state of n_worlds identical worlds on `device` and reset them to the model's
 * "neutral" keyframe.  Replaces: mjw.put_data(nworld=...)  (warp/simulation.py:418-424).
 * NOT stream-ordered (allocations, uploads, one synchronising reset).
 * A model compiled with option/noslip_iterations > 0 (the reference's CPU class: mujoco_globals.yaml:15 under mujoco.mj_step,
 * simulation.py:74-76) runs that friction-only post-pass on every step in contact — the engine of flygym_amd.Simulation, meant for
 * ONE world: the batch then also allocates the pass's scratch, 198 x 200 floats = 158 400 bytes PER WORLD (4096 worlds: 649 MB;
 * nmf_batch_info out[12] says whether), and a step that takes the primal path costs one articulated-body solve per constraint row more.  The
 * reference's batched class strips the option before it compiles the model (warp/simulation.py:427-448) and so does
 * flygym_amd.HIPSimulation; a C caller that wants the batched semantics compiles its model with noslip_iterations = 0. */
nmf_batch* nmf_batch_create(const nmf_model* model, int n_worlds, int device);
void nmf_batch_destroy(nmf_batch* batch);
int nmf_batch_n_worlds(const nmf_batch* batch);

/* The same with explicit options (NULL = defaults = nmf_batch_create).  Every field: 0 = the library's default.  These are the
 * switches the tests and A/B measurements use; none changes WHAT is computed beyond solver tolerance (`solver`) — `sched`,
 * `order`, `max_chunks`, `chunk_div`, `min_chunk_steps`, `order_every` never change a result (worlds are independent).
 * The library reads NO environment variable unless NMF_ALLOW_ENV=1 is set in the process environment (development: the
 * NMF_SOLVER / NMF_SCHED / NMF_ORDER / NMF_MAX_CHUNKS / NMF_CHUNK_DIV / NMF_MIN_CHUNK_STEPS / NMF_ORDER_EVERY /
 * NMF_DISABLE_REST_FAST variables then override these options); what a batch actually runs is reported by nmf_batch_info. */
typedef struct nmf_batch_options {
  int32_t struct_size;      /* sizeof(nmf_batch_options) as the caller was compiled                                            */
  int32_t solver;           /* bit 0: every step on the primal Newton loop; bit 1: the contact-space solve starts from the start
                               point's own sign pattern, not from the previous step's active set; bit 2: its ends are never
                               re-solved on the primal loop (no residual test)                                                 */
  int32_t sched;            /* 1: whole-launch work items (no chunks)                                                          */
  int32_t order;            /* world order of over-subscribed launches: 1 in index order (order kernel), 2 costliest first,
                               3 none (no order kernel), 4 the measured policy of rounds 1-2                                    */
  int32_t max_chunks;       /* 1..16                                                                                           */
  int32_t min_chunk_steps;  /* >= 1                                                                                            */
  int32_t order_every;      /* costliest-first order recomputed every n-th launch                                              */
  int32_t rest_slow;        /* 1: hybrid kernels use the table-driven level passes of the rest of the body                     */
  float chunk_div;          /* each chunk takes 1 / chunk_div of the steps that are left (> 1)                                 */
  int32_t flies_per_cu;     /* > 0: at most this many of the stepping kernel's single-wave workgroups per CU (the kernel's own
                               residency is the ceiling): the launch carries that much idle LDS per workgroup, which leaves LDS and
                               vector registers of every CU to a kernel of ANOTHER stream — the eye renderer of the previous vision
                               tick beside the next tick's physics (co-residency)                          */
} nmf_batch_options;
nmf_batch* nmf_batch_create_ex(const nmf_model* model, int n_worlds, int device, const nmf_batch_options* options);

/* What the batch runs — so that a result can always be traced to the code that made it:
 * out[0] kernel family (0 LEGS_ONLY star, 1 LEGS_ACTIVE_ONLY star, 2 / 3 general tree 144 / 216 dofs, 4 ALL_BIOLOGICAL hybrid,
 * 5 ALL_POSSIBLE hybrid), out[1] terrain kernel, out[2] tether (weld) kernel, out[3] contact-space solve (0 none: primal loop
 * only, 1 star flavour, 2 hybrid flavour), out[4] contacts it takes (steps with more go to the primal loop), out[5] flies
 * (workgroups) per CU, out[6] resident workgroups of the device, out[7] chunked launches (0 / 1), out[8] max chunks,
 * out[9] 1000 * chunk_div, out[10] world-order policy (0 in order, 1 costliest first, 2 none, 3 auto, -1 measured), out[11]
 * solver option bits in effect, out[12] noslip iterations (> 0: the CPU class's flavour — the batch holds 158 400 bytes of noslip
 * scratch per world, see nmf_batch_create), out[13] contact capacity, out[14] LDS bytes / out[15] vector registers of the
 * stepping kernel (from the loaded code object). */
int nmf_batch_info(const nmf_batch* batch, int32_t out[16]);

/* Contacts kept per world and step: min(max_contacts, 48); returns the capacity in effect (negative on error).  Contacts
 * beyond it, in geom order, are dropped and the step counts as overflowed (stats column 2, nmf_stats overflow steps).
 * Call between launches (synchronous).  Replaces: the nconmax of mjw.put_data  (warp/simulation.py:50-56, 418-424). */
int nmf_batch_set_contact_capacity(nmf_batch* batch, int max_contacts);

/* Reset every world to the neutral keyframe (qpos, ctrl from the keyframe; qvel = 0; time = 0)
 * and refresh the pose outputs.  Replaces GPUSimulation.reset (warp/simulation.py:64-71). */
int nmf_reset(nmf_batch* batch, void* stream);

/* Same for the worlds with mask_dev[w] != 0 only (uint8 per world, device memory); the others keep their state and
 * their own clock.  No reference counterpart (its reset re-uploads every world): episode resets for RL loops. */
int nmf_reset_worlds(nmf_batch* batch, const uint8_t* mask_dev, void* stream);

/* Advance all worlds by n_steps physics steps in ONE kernel launch, controls held constant.
 * Replaces n_steps calls of GPUSimulation.step (warp/simulation.py:260-263). */
int nmf_step(nmf_batch* batch, int n_steps, void* stream);

/* Same, but before step s the controls ctrl[w][act_ids[a]] are loaded from
 * table[w][(start + s) % table_steps][a]  (a < n_act): the device-resident kinematic-replay
 * loop of the reference benchmark (src/flygym_demo/benchmark/time_gpu_simulation.py:137-150:
 * update_target_angles_kernel + set_actuator_inputs + step + increment_counter, graph-captured). */
int nmf_step_replay(nmf_batch* batch, const float* table_dev, int table_steps, int n_act,
                    const int32_t* act_ids_dev, int start, int n_steps, void* stream);

/* The same launch (table_dev NULL: nmf_step, controls held; else nmf_step_replay with n_act_table columns) that also RECORDS the
 * observation block after every obs_every-th step — what the reference's loops read after every step (get_joint_angles /
 * get_joint_velocities / get_actuator_forces / get_ground_contact_info, src/flygym/simulation.py:142-243; MJWarp computes its
 * sensors every step, warp/simulation.py:260-263), without leaving the kernel:
 *   ring[(s + 1) / obs_every - 1][w][0 .. 2 n_joint + n_act + 96) for every step s (0-based) with (s + 1) % obs_every == 0,
 * in the layout of nmf_pack_observations — [joint angles | joint velocities | forces of the first n_act actuators | the 96
 * contact-sensor floats] — and with its values: each row is bit for bit what nmf_pack_observations gives when the launch ends at
 * that step.  ring_dev: float32 [n_steps / obs_every][n_worlds][row_stride].  The contact sensors and actuator forces are
 * evaluated on the recorded steps (and, as always, on the launch's last step for the batch's own arrays).
 * n_steps must be a multiple of obs_every (a trailing partial window would be stepped but never recorded: refused). */
int nmf_step_record(nmf_batch* batch, const float* table_dev, int table_steps, int n_act_table, const int32_t* act_ids_dev, int start,
                    int n_steps, int obs_every, int n_joint, int n_act, float* ring_dev, int row_stride, void* stream);

/* Device pointer + row width of a per-world field (zero-copy views for PyTorch). */
float* nmf_field_ptr(nmf_batch* batch, int field, int32_t* width);

/* dst[w][k] = field[w][ids[k]*group .. +group)  — gather in caller order (group = 1, 3 or 4).
 * Replaces wp_gather_indexed_cols_2d / _rows_vec3f / _rows_quatf (warp/utils.py:29-127). */
int nmf_gather(nmf_batch* batch, int field, const int32_t* ids_dev, int n_ids, int group,
               float* dst_dev, void* stream);

/* field[w][ids[k]] = src[w][k]  — scatter controls in caller order.
 * Replaces wp_scatter_indexed_cols_2d (warp/utils.py:84-104). */
int nmf_scatter(nmf_batch* batch, int field, const int32_t* ids_dev, int n_ids,
                const float* src_dev, void* stream);

/* The observation block of the multi-GPU exchange, packed in ONE launch: out[w][0 .. 2 n_joint + n_act + 96) =
 * [joint angles (qpos 7 ..) | joint velocities (qvel 6 ..) | forces of the first n_act actuators | the 96 contact-sensor
 * floats], rows row_stride floats apart.  Replaces the four getter launches the reference would need per tick
 * (get_joint_angles / get_joint_velocities / get_actuator_forces / contact sensors, warp/simulation.py:73-211) as the
 * input of the RCCL all-gather (flygym_amd.sharding.ObsGather; the reference is single-GPU, warp/utils.py:192-202). */
int nmf_pack_observations(nmf_batch* batch, int n_joint, int n_act, float* out_dev, int row_stride, void* stream);

/* Number of physics steps taken since the last reset (host-side counter, no sync). */
int64_t nmf_step_count(const nmf_batch* batch);

/* Timing helper for benchmarks: runs nmf_step/nmf_step_replay `reps` times on `stream`
 * bracketed by hipEvents recorded on that same stream; returns mean milliseconds per launch
 * (blocks the host).  table_dev may be NULL for constant controls. */
double nmf_time_launches(nmf_batch* batch, const float* table_dev, int table_steps, int n_act,
                         const int32_t* act_ids_dev, int n_steps, int reps, void* stream);

/* Measurement helper: the shader clock the stepping kernel actually ran at.  Workgroup 0 of every stepping launch reads
 * the shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) when it starts and when it
 * ends; the library accumulates both differences.  *hz_out = 1e8 * (shader cycles) / (100 MHz ticks) over the launches
 * since the last call with reset != 0 (0 if there were none).  Synchronises with the device.  No reference
 * counterpart: bench.py prices the kernel's instruction issue against the clock of the run it measures. */
int nmf_shader_clock(nmf_batch* batch, double* hz_out, int reset);

/* ---- sensors the north star names; the reference snapshot holds only their constants
 * (src/flygym/assets/model/legacy/flygym1_config.yaml:141-192), so semantics are build-defined. ---- */

/* Hex-ommatidia resample of raw eye images.  images[n_images][n_pixels][3] uint8 RGB; id_map[n_pixels] 16-bit:
 * bits 0..14 = 0 (no ommatidium) or k (ommatidium k-1), bit 15 = that ommatidium is pale; shared by all images;
 * pale[n_ommatidia] uint8 (1 = pale type, reads blue; 0 = yellow type, reads green); inv_norm[k] = 1 / (255 * pixels of ommatidium k);
 * out[n_images][n_ommatidia][2] float32 (channel 0 yellow, 1 pale).  Buffers 16-byte aligned.
 * plan: NULL, or the run plan of the id map written by nmf_retina_plan into a buffer of nmf_retina_plan_bytes(n_pixels)
 * bytes (16 bytes per 16-pixel chunk, then the list of chunks that touch an ommatidium): with a plan
 * and n_pixels a multiple of 1024 the frames are streamed with fully coalesced loads; the results
 * are identical either way (integer sums). */
size_t nmf_retina_plan_bytes(int n_pixels);
int nmf_retina_plan(const int16_t* id_map_dev, int n_pixels, void* plan_dev, void* stream);
int nmf_retina_resample(const uint8_t* images_dev, const int16_t* id_map_dev, const void* plan_dev, const uint8_t* pale_dev,
                        const float* inv_norm_dev, int n_images, int n_pixels, int n_ommatidia,
                        float* out_dev, void* stream);

/* Compound-eye renderer fused with the ommatidia resample (no reference counterpart in this snapshot: the eye cameras,
 * like the resample, survive only as constants in src/flygym/assets/model/legacy/flygym1_config.yaml:141-173; the
 * reference's image path is rendering.py / warp/rendering.py -> MuJoCo / MJWarp renderers).  Build-defined (DESIGN.md
 * section 7): each eye is an equidistant-fisheye camera attached to a body segment; the scene is the ground (checker
 * plane, or the terrain relief of the batch's world), a uniform sky, up to 8 spheres and the fly's own body as up to 64
 * capsules rigidly attached to segments (what reference warp/rendering.py:385-441 gives the batch renderer: the whole
 * model).  Reads the segment poses of the last nmf_step / nmf_reset. */
typedef struct nmf_eye_params {
  int32_t height, width;        /* raw frame size in pixels; height * width a multiple of 16                           */
  float fov_deg;                /* full angle seen along the vertical image axis (ray angle is proportional to radius)   */
  int32_t eye_seg[2];           /* parent segment of each eye camera (index into the batch's segment order)             */
  float rel_pos[2][3];          /* camera position in the parent segment frame                                           */
  float rel_quat[2][4];         /* camera orientation in the parent frame (w,x,y,z); the camera looks along -z, +y is up */
  float checker_size;           /* side of a ground checker square                                                       */
  uint8_t sky_rgb[4], ground_rgb[2][4], sphere_rgb[8][4];   /* colours (4th byte unused)                                */
  int32_t n_spheres;            /* 0..8                                                                                  */
  int32_t spheres_per_world;    /* 1: spheres_dev is [n_worlds][n_spheres][4]; 0: [n_spheres][4] shared by all worlds   */
  uint8_t wall_rgb[4], body_rgb[4];   /* side walls of the terrain relief; the fly's own body                          */
  int32_t n_capsules;           /* 0..64 capsules of the fly's own body (capsule_seg_dev / capsule_geom_dev)            */
  int32_t terrain_relief;       /* 1: the ground is the height map the batch's physics collides with (gapped / blocks /
                                   mixed worlds: constant-height cells with side walls); 0: the flat checker plane      */
  int32_t rays_per_ommatidium;  /* 0: every pixel of the raw frame that lies in an ommatidium's cell is cast (readings =
                                   resampling the rendered frame, bit for bit); 16: sixteen of each cell's pixels — at
                                   indices floor((2 j + 1) n / 32) of its n pixels in raster order — and the reading is
                                   their mean: 15 x fewer rays, an approximation of the cell mean (no frames in this mode) */
} nmf_eye_params;

/* sizeof(nmf_eye_params) as this library was compiled: lets a foreign-language binding verify its struct layout. */
size_t nmf_eye_params_size(void);

/* The visit plan of the renderer: which 16-pixel chunks feed an ommatidium, in which order, with the bounding cones of their rays,
 * the sampled mode's pixel lists — and the plan's OWN device copies of the id map, the retina run plan (nmf_retina_plan), the pale
 * flags and the normalisation, so that later changes to the caller's buffers cannot reach a render.  nmf_eye_plan_create is NOT
 * stream-ordered and must not be called inside a stream capture: it synchronises the device, copies the id map / run plan / pale
 * flags to the host, sorts there, allocates and uploads (tens of milliseconds, once per id map and lens).  The plan belongs to the
 * device it was made on and to no batch: any batch on that device can render with it.  NULL on error (nmf_last_error). */
nmf_eye_plan* nmf_eye_plan_create(const int16_t* id_map_dev, const void* plan_dev, const uint8_t* pale_dev, const float* inv_norm_dev,
                                  int height, int width, float fov_deg, int n_ommatidia, int device);
void nmf_eye_plan_destroy(nmf_eye_plan* plan);

/* Render with an explicit plan: argument checks and ONE kernel launch — stream-ordered, hipGraph-capturable (a vision tick =
 * nmf_step + nmf_eye_render_planned captures as one graph).  params->height / width / fov_deg must be the plan's. */
int nmf_eye_render_planned(nmf_batch* batch, const nmf_eye_params* params, const nmf_eye_plan* plan, const float* spheres_dev,
                           const int32_t* capsule_seg_dev, const float* capsule_geom_dev, uint8_t* frames_out_dev, float* omm_out_dev,
                           void* stream);

/* The same without a handle: the batch builds a plan on the first call with a new (id_map_dev, plan_dev, pale_dev, inv_norm_dev,
 * shape, lens) — keyed on the buffer ADDRESSES — and keeps up to four of them (least recently used first out).  That first call
 * is therefore NOT stream-ordered (see nmf_eye_plan_create) and not capturable; later calls with the same key are.  The plan holds
 * copies: a caller that rewrites one of the four buffers in place keeps rendering with the old contents until it passes other
 * addresses — use the explicit handle where that matters.
 * spheres_dev: (x, y, z, radius) per sphere, float32.  id_map / plan / pale / inv_norm as for nmf_retina_resample (the
 * plan is required).  frames_out_dev: NULL or uint8 [n_worlds][2][height*width][3] raw eye frames;
 * omm_out_dev: NULL or float32 [n_worlds][2][n_ommatidia][2].  Rendering frames and resampling them with
 * nmf_retina_resample gives bit-identical ommatidia readings (integer sums).
 * capsule_seg_dev[n_capsules] int32 segment index, capsule_geom_dev[n_capsules][7] float32 (end points p0, p1 in the
 * segment frame, radius) — NULL when n_capsules is 0. */
int nmf_eye_render(nmf_batch* batch, const nmf_eye_params* params, const float* spheres_dev,
                   const int32_t* capsule_seg_dev, const float* capsule_geom_dev, const int16_t* id_map_dev, const void* plan_dev, const uint8_t* pale_dev, const float* inv_norm_dev,
                   int n_ommatidia, uint8_t* frames_out_dev, float* omm_out_dev, void* stream);

/* Odor intensity at n_sensors points rigidly attached to named segments (sensor_seg = index into the
 * batch's segment order, sensor_rel = offset in the segment frame): out[w][d][k] = sum_s peak[s][d] / dist^2.
 * Reads the pose outputs of the last nmf_step / nmf_reset. */
int nmf_odor_intensity(nmf_batch* batch, const int32_t* sensor_seg_dev, const float* sensor_rel_dev, int n_sensors,
                       const float* source_pos_dev, const float* source_peak_dev, int n_sources, int n_dims,
                       float* out_dev, void* stream);

/* Kinematic-replay data path on the device.  Replaces MotionSnippet.get_joint_angles (reference
 * src/flygym_demo/spotlight_data/preprocessing.py:80-142: scipy savgol_filter + interp1d(kind="cubic") on the host):
 * Savitzky-Golay smoothing of the recorded joint angles, the not-a-knot cubic spline through the smoothed frames, its
 * values at t = k * out_dt (k < n_out; the last frame's value beyond the last knot), in float64, stored as float32.
 * clip_dev[n_frames][n_cols] float32; out_dev[n_out][n_cols] float32;
 * sg_taps_dev (float64, device): window interior taps, then window/2 rows of `window` taps for the first window/2 frames
 * (polynomial fit over the first window), then window/2 rows for the last window/2 frames — the constants of
 * savgol_filter(mode="interp"), computed by the caller (flygym_amd.replay.savgol_taps). */
int nmf_replay_resample(const float* clip_dev, int n_frames, int n_cols, double fps, double out_dt,
                        const double* sg_taps_dev, int window, int n_out, float* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NMF_H_ */
