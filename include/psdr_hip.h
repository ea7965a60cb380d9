/*
 * psdr_hip.h -- C ABI of the MI355X-native differentiable wavefront renderer.
 *
 * This is the drop-in boundary for the hot path of uci-rendering/psdr-cuda
 * (Integrator.renderC / renderD and everything below it).  The reference has
 * no C ABI of its own: its boundary is the pybind11 module `psdr_cuda`
 * (reference src/psdr.cpp:40-295).  Every entry point below names the
 * reference interface it replaces (file:line relative to the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success, non-zero on failure;
 *     psdr_last_error() returns the message (the reference throws
 *     psdr::Exception : std::runtime_error, include/misc/Exception.h:86-132).
 *   - all table pointers are CALLER-OWNED.  For libpsdr_hip.so they are
 *     DEVICE pointers (HBM); for the CPU oracle (oracle/psdr_oracle.cpp, test
 *     infrastructure only) the same structs carry HOST pointers.
 *   - float = IEEE fp32, indices = int32, images are interleaved RGB
 *     [H*W][3], pixel i = y*W + x, row 0 = top (reference
 *     src/integrator/integrator.cpp:76-88, docs/python_render.rst).
 *
 * The AD graph of the reference (Enoki DiffArray) is cut at the "scene
 * tables" built by Scene::configure (reference src/scene/scene.cpp:56-278):
 * kernels consume the tables (+ tangent tables in forward mode) and produce
 * the image (+ derivative image), or consume an adjoint image and scatter-add
 * into gradient tables (reverse mode).
 */
#ifndef PSDR_HIP_H
#define PSDR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- table strides (in 4-byte words) ------------------------------------ */
#define PSDR_TRI_STRIDE   24 /* p0 e1 e2 n0 n1 n2 face_normal (7x3) face_area, 2 pad
                                = TriangleInfo_, include/psdr/types.h:135-146 */
#define PSDR_TRIUV_STRIDE  8 /* uv0 uv1 uv2 (3x2), 2 pad = TriangleUV, types.h:154-155 */
#define PSDR_SEDGE_STRIDE 16 /* p0 e1 n0 n1 p2 (5x3) is_boundary(0/1)
                                = SecondaryEdgeInfo_, include/psdr/edge/edge.h:49-65 */
#define PSDR_PEDGE_STRIDE  8 /* p0(2) p1(2) edge_normal(2) edge_length, 1 pad
                                = PrimaryEdgeInfo_, edge.h:27-40 */
#define PSDR_BSDF_STRIDE  16 /* int32: type, then 5 x (texel offset, width, height) */
#define PSDR_EMITTER_F_STRIDE 8 /* radiance rgb, sampling_weight (normalised),
                                   inv_total_area, face pmf sum, 2 pad
                                   (src/emitter/area.cpp:10-62, src/shape/mesh.cpp:239-249) */
#define PSDR_EMITTER_I_STRIDE 4 /* mesh id, first global tri id, num faces, offset into face_cmf/pmf */
#define PSDR_CAM_WORDS    64 /* see below */
#define PSDR_ENV_WORDS    32 /* EnvironmentMap record (src/emitter/envmap.cpp:10-143,
                                include/psdr/emitter/envmap.h:38-46), see PSDR_ENV_* below */

/* tri_mesh[t] = mesh id | PSDR_TRI_FACE_NORMALS if the mesh uses face normals
   (scene.cpp:205-216 m_triangle_face_normals) */
#define PSDR_TRI_FACE_NORMALS 0x40000000

/* BSDF types (src/bsdf/diffuse.cpp, src/bsdf/roughconductor.cpp) */
#define PSDR_BSDF_DIFFUSE        0
#define PSDR_BSDF_ROUGHCONDUCTOR 1
/* bsdf_rec parameter slots: slot s occupies words 1+3*s .. 3+3*s */
#define PSDR_SLOT_REFLECTANCE 0 /* Diffuse::m_reflectance / RoughConductor::m_specular_reflectance (3 ch) */
#define PSDR_SLOT_ALPHA_U     1 /* 1 ch */
#define PSDR_SLOT_ALPHA_V     2 /* 1 ch */
#define PSDR_SLOT_ETA         3 /* 3 ch */
#define PSDR_SLOT_K           4 /* 3 ch */

/* cam[] layout (PerspectiveCamera, src/sensor/perspective.cpp:11-33), row-major 4x4 */
#define PSDR_CAM_SAMPLE_TO_CAMERA  0
#define PSDR_CAM_TO_WORLD         16
#define PSDR_CAM_WORLD_TO_SAMPLE  32
#define PSDR_CAM_POS              48
#define PSDR_CAM_DIR              51
#define PSDR_CAM_INV_AREA         54

/* env_f[] layout (EnvironmentMap): m_from_world and m_to_world as row-major 3x3 blocks (only
   transform_dir is ever applied, include/psdr/core/transform.h:91-94), m_scale, and the
   scene AABB the sampled directions are projected on (m_lower / m_upper, scene.cpp:135-141) */
#define PSDR_ENV_FROM_WORLD  0
#define PSDR_ENV_TO_WORLD    9
#define PSDR_ENV_SCALE      18
#define PSDR_ENV_LOWER      19
#define PSDR_ENV_UPPER      22

/* integrators (src/psdr.cpp:282-294; PathTracer is build-defined) */
#define PSDR_INTEGRATOR_DIRECT 0
#define PSDR_INTEGRATOR_PATH   1
#define PSDR_INTEGRATOR_FIELD  2
/* FieldExtractionIntegrator fields (src/integrator/field.cpp:10-54) */
#define PSDR_FIELD_SILHOUETTE 0
#define PSDR_FIELD_POSITION   1
#define PSDR_FIELD_DEPTH      2
#define PSDR_FIELD_GEONORMAL  3
#define PSDR_FIELD_SHNORMAL   4
#define PSDR_FIELD_UV         5

/* Scene tables = everything Scene::configure() leaves on the device
   (reference src/scene/scene.cpp:56-278). */
typedef struct psdr_scene_desc {
    int32_t width, height;               /* RenderOption, types.h:171-182 */
    int32_t num_tris, num_meshes, num_bsdfs, num_emitters;
    int32_t num_sec_edges, num_prim_edges, num_texels, num_guide_cells;

    const float   *tri_info;             /* [T][PSDR_TRI_STRIDE]   scene.cpp:205-216; MUST be 16-byte aligned (rows are staged as float4) */
    const float   *tri_uv;               /* [T][PSDR_TRIUV_STRIDE] or NULL (all zero) */
    const int32_t *tri_mesh;             /* [T] */
    const int32_t *mesh_bsdf;            /* [M] bsdf id (Mesh::m_bsdf); -1 = the bounding mesh of the
                                            environment map (scene.cpp:135-172, bsdf == nullptr) */
    const int32_t *mesh_emitter;         /* [M] emitter id or -1 (Mesh::m_emitter) */
    const int32_t *bsdf_rec;             /* [Nb][PSDR_BSDF_STRIDE] */
    const float   *texels;               /* pool of Bitmap data; 3-ch textures interleaved RGB */
    const float   *emitter_f;            /* [Ne][PSDR_EMITTER_F_STRIDE] */
    const int32_t *emitter_i;            /* [Ne][PSDR_EMITTER_I_STRIDE] */
    const float   *face_cmf, *face_pmf;  /* per-emitter-mesh face distributions, mesh.cpp:248-249 */
    const float   *emitter_cmf, *emitter_pmf; /* [Ne] Scene::m_emitters_distrb, scene.cpp:183-196 */
    float          emitter_sum;
    const float   *cam;                  /* [PSDR_CAM_WORDS] */
    const float   *sec_edge;             /* [E][PSDR_SEDGE_STRIDE] scene.cpp:219-244 */
    const float   *sec_cmf, *sec_pmf;    /* [E] pmf = |e1| */
    float          sec_sum;
    const float   *prim_edge;            /* [Ep][PSDR_PEDGE_STRIDE] perspective.cpp:39-111 */
    const float   *prim_cmf, *prim_pmf;  /* [Ep] pmf = screen length */
    float          prim_sum;
    int32_t        guide_reso[3];        /* HyperCubeDistribution3f, src/core/cube_distrb.cpp */
    const float   *guide_cmf, *guide_pmf;/* [num_guide_cells] or NULL (no guiding) */
    float          guide_sum;
    /* EnvironmentMap (at most one, scene_loader.cpp:294-311): env_emitter = its index in the emitter
       tables or -1; its emitter_i row names the bounding mesh (scene.cpp:135-172).  The lat-long
       radiance bitmap lives in the texel pool (so d_texels / g_texels cover it). */
    int32_t        env_emitter;
    int32_t        env_tex[3];           /* texel offset, width, height of m_radiance */
    int32_t        env_reso[2];          /* m_cell_distrb resolution ((w-1)*2, (h-1)*2), envmap.cpp:14-15 */
    const float   *env_f;                /* [PSDR_ENV_WORDS] */
    const float   *env_cmf, *env_pmf;    /* [env_reso[0]*env_reso[1]] luminance*sin(theta), envmap.cpp:17-21 */
    float          env_sum;
    /* Bit t set = a BSDF of type t (PSDR_BSDF_*) occurs in bsdf_rec; 0 = unknown (the library then keeps
       the code of every BSDF type in its kernels).  A host that knows its materials sets it so that, e.g.,
       an all-diffuse scene runs the kernel variant compiled without the GGX / conductor-Fresnel code.
       Bits that are clear MUST be right: a cleared type is evaluated as diffuse. */
    uint32_t       material_mask;
    /* [E][2] global triangle ids of the one or two faces adjacent to every secondary edge (second = -1 on a
       boundary edge), or NULL.  Not in the reference's SecondaryEdgeInfo (edge.h:27-65): the two rays that
       eval_secondary_edge (direct.cpp:246-254) starts ON the edge cannot hit these faces again in exact
       arithmetic (a line meets a plane once, at the edge point); in fp32 the rounded edge point lets a grazing
       ray re-hit them just above RayEpsilon -- ~3e-3 of the boundary term.  With the table the two rays skip
       the adjacent faces; without it they are traced as the reference does. */
    const int32_t *sec_edge_faces;
    /* [Ep][4] per primary edge: 1 / depth along the viewing direction (dot(vertex - camera, normalize(cam[PSDR_CAM_DIR])))
       of its two end points, then the global ids of its adjacent faces as int32 bit patterns (second = first on a boundary
       edge); or NULL.  When present the primary-edge term runs the reference's PSDR_PRIMARY_EDGE_VIS_CHECK variant
       (macros.h:13, integrator.cpp:105-108, perspective.cpp:91-96,171-196): an edge sample counts only if the edge point
       itself is visible from the camera (the ray through it with tmax = distance - 100 ShadowEpsilon hits nothing).  The
       reference keeps the sample-space z of the end points and unprojects the interpolated point; 1 / depth is affine along
       the film segment in the same way and distance = depth / cos is the same number in exact arithmetic, but
       sample-space z = far (1 - near / depth) / (far - near) holds only 3-4 digits of the depth in fp32 (near 0.1, depth
       500: +-0.15 units against the 1e-3 margin), and the edge's own faces -- met exactly AT the distance -- are skipped
       rather than left to round-off. */
    const float   *prim_edge_z;
} psdr_scene_desc;

/* psdr_render_opts.flags: execution strategy of the PathTracer interior term.
   default: the library chooses.  FUSED = one lane carries a whole path in registers.
   WAVEFRONT = one kernel per bounce over SoA path-state streams in HBM with wave-ballot stream
   compaction of the live paths between bounces (renderC and material-only renderD). */
#define PSDR_FLAG_FUSED     1
#define PSDR_FLAG_WAVEFRONT 2
/* Evaluate the three expressions the library replaces by fp32-robust equivalents in the reference's LITERAL
   form instead: the solid-angle hit point as p = ray(t) (scene.cpp:368), the rays that start on a secondary edge without skipping
   the adjacent faces (direct.cpp:246-254; as if sec_edge_faces were NULL), and "the camera sees p1" as the fp32 distance
   |its1.p - p1| < ShadowEpsilon (direct.cpp:262).  psdr_render_c / psdr_render_d_fwd only (reverse mode has no literal-form adjoint):
   makes the difference between the two forms measurable on the device against the oracle's reference_form. */
#define PSDR_FLAG_LITERAL_FORMS 4
/* psdr_render_c only: the caller is going to differentiate THIS render in reverse mode (renderD + enoki.backward: the primal image first, the adjoint image
   later).  Where the following psdr_render_d_rev would run a split launch whose value sweep is exactly this render -- PathTracer on a two-level scene, the
   traced wavefront, one chunk of slots -- the render runs with recording stages and the per-path records stay on the handle; a psdr_render_d_rev with the
   same options, the same tables (psdr_scene_set_tables with an identical descriptor in between is fine; the caller must not have overwritten them in
   place) and out_img = NULL then runs its adjoint kernel only (C4 shard: 15.6 of its 34.5 ms).  Where the wavefront does not apply (a scene without a
   tree: the 12-triangle cbox) the PathTracer render runs as the value kernel of a split reverse launch instead (image + one record per path) and a
   psdr_render_d_rev that asks for a geometry table (tri_info / cam_to_world) runs the adjoint kernel on it (cbox 512^2 x 64: 0.85 + 4.86 -> 1.25 + 4.0 ms).
   Ignored wherever neither applies (other integrators, paths deeper than 8, more slots than one chunk, option rev_split = 0). */
#define PSDR_FLAG_KEEP_RECORDS 8

/* One render call = Integrator::renderC / renderD on one shard of the sample
   slots (src/integrator/integrator.cpp:13-119, src/integrator/direct.cpp). */
typedef struct psdr_render_opts {
    int32_t integrator;                  /* PSDR_INTEGRATOR_* */
    int32_t bsdf_samples, light_samples; /* DirectIntegrator ctor, direct.cpp:32-34 */
    int32_t max_depth;                   /* PathTracer; psdr_render_d_rev: up to 8 the per-lane path record lives in LDS, up to 250 in HBM */
    int32_t hide_emitters;               /* DirectIntegrator::m_hide_emitters */
    int32_t field;                       /* PSDR_FIELD_* */
    int32_t spp, sppe, sppse;            /* GLOBAL counts: normalisation + RNG stream index */
    int32_t spp_begin, spp_end;          /* this call evaluates s in [begin,end) of every pixel */
    int32_t sppe_begin, sppe_end;        /* slots [W*H*begin, W*H*end) of sampler 1 */
    int32_t sppse_begin, sppse_end;      /* slots [W*H*begin, W*H*end) of sampler 2 */
    int32_t flags;                       /* PSDR_FLAG_* (0 = library chooses) */
    uint64_t rng_offset[3];              /* draws already consumed per stream of sampler 0/1/2:
                                            the reference keeps the PCG32 states alive across
                                            render calls (scene.cpp:65-79) */
} psdr_render_opts;

/* Forward-mode tangent tables (d table / d P); any pointer may be NULL (= 0). */
typedef struct psdr_tangents {
    const float *d_tri_info;             /* [T][PSDR_TRI_STRIDE] */
    const float *d_texels;               /* [num_texels] */
    const float *d_emitter_rad;          /* [Ne][3] */
    const float *d_cam_to_world;         /* [16] row-major */
    const float *d_sec_edge;             /* [E][PSDR_SEDGE_STRIDE] (p0, e1 used... all 15) */
    const float *d_prim_edge;            /* [Ep][PSDR_PEDGE_STRIDE] (p0, p1 used) */
    const float *d_env_f;                /* [PSDR_ENV_WORDS] (from_world and scale used) */
} psdr_tangents;

/* Reverse-mode gradient tables (accumulated with +=); NULL = not wanted. */
typedef struct psdr_grads {
    float *g_tri_info;
    float *g_texels;
    float *g_emitter_rad;
    float *g_cam_to_world;
    float *g_sec_edge;
    float *g_prim_edge;
    float *g_env_f;
} psdr_grads;

typedef struct psdr_scene_s *psdr_scene_t;

/* error text of the last failing call on this thread */
const char *psdr_last_error(void);
/* "psdr-hip <version> gfx950" */
const char *psdr_version(void);
/* sizeof(psdr_scene_desc), sizeof(psdr_render_opts), sizeof(psdr_tangents), sizeof(psdr_grads):
   lets a foreign-language binding verify its struct mirrors before the first call */
int psdr_abi_struct_sizes(int32_t out[4]);

/* Scene handle; replaces Scene() / ~Scene() (src/scene/scene.cpp:19-41). */
int psdr_scene_create(psdr_scene_t *out);
int psdr_scene_destroy(psdr_scene_t h);
/* Developer options of a handle -- the A/B switches of tools and tests (the reference has none: its strategies are fixed by OptiX and
   Enoki); the library reads NO environment variable.  Names (value): bvh_refit, tiny_scene, two_level, wf_binned, wf_traced, sort_edges,
   tiny_variants, sink_private, aa_prims (0 / 1); bvh_build (1 device, 0 host, -1 by size); wide (0: never the 4-wide tree in the render kernels);
   rev_split, sedge_split (1 / 0 force, -1 default rule); keep_records (0: PSDR_FLAG_KEEP_RECORDS is ignored); logd (0: PathTracer forward mode with tangents on
   diffuse albedo texels only runs the dual-number kernel, never the log-derivative one); rev_sorted (0: reverse PathTracer kernels
   scatter row adjoints on the spot); wf_geo (0: PathTracer geometry tangents through the fused kernel); tangent_live (0: no liveness mask); forest_min_inline (inline triangles a two-level tree needs at least, default 6); scratch_plain (1: scratch blocks of 64 MB and more come from hipMalloc instead of the stream-ordered pool -- process-wide, an experiment on TLB reach);
   own_pixels (0: the camera kernels always add to the image with atomics, also where one wave holds all samples of a pixel); emitter_pretest (0: BSDF-sampled rays whose hit matters only on an emitter -- DirectIntegrator, a PathTracer path's last vertex -- are traced without
   first meeting the emitters' primitives); occ_rows (0: the light rays of a scene without a tree test every kernel-argument primitive instead of
   the ones that can lie between the path vertex and the emitter sample); probe (0: no probe / trace / final launches);
   trace_wg2 (dense trace kernel as two workgroups per CU: -1 by forest and launch size, 0 never, n > 0 always with stack columns of n entries);
   chunk_log2 (slots per chunk of the chunked launches, 0 = default); blocks_per_cu, camera_blocks, lds_budget, sink_rep, bvh_maxleaf (integers,
   0 = default where that makes sense); bvh_tcost (float).  Unknown names fail.  Options that change the tree take effect at the next psdr_bvh_build. */
int psdr_scene_set_option(psdr_scene_t h, const char *name, double value);

/* Replaces the table-publishing tail of Scene::configure (scene.cpp:199-244):
   remembers the caller-owned device tables. */
int psdr_scene_set_tables(psdr_scene_t h, const psdr_scene_desc *desc);

/* Replaces Scene_OptiX::configure + the OptiX GAS build
   (src/scene/scene_optix.cpp:34-72, include/psdr/scene/optix.h:277-340):
   builds the BVH over tri_info (p0,e1,e2) of the current tables.  When the triangle count is
   unchanged since the last build (an optimisation loop moving vertices) the tree is refitted on
   the device instead of rebuilt (OptiX's build is a device build too); it is rebuilt when the
   refitted boxes have grown by 30 % in area or after 64 refits.  psdr_scene_set_option(h, "bvh_refit", 0)
   forces a rebuild every time (the library reads no environment variable). */
int psdr_bvh_build(psdr_scene_t h, void *stream);
/* out = { full builds, refits, inner nodes, tree depth } of this handle (diagnostics) */
int psdr_bvh_stats(psdr_scene_t h, int32_t out[4]);
/* out = { primitives carried in the kernel arguments (triangles or parallelograms), trees of a two-level scene,
   inline triangles of a two-level scene, leaf triangles in the tree(s), 1 if the tree was built on the device,
   how many of [0] are axis-aligned rectangles in slab form, 1 if the light rays of a scene without a tree test occluder rows only, the most rows any of
   those entries names } -- what psdr_bvh_build chose for the current tables (diagnostics; bench.py prices its
   per-ray arithmetic floor from [0]).  No reference counterpart: OptiX hides its acceleration structure. */
int psdr_scene_info(psdr_scene_t h, int32_t out[8]);

/* Replaces Scene_OptiX::ray_intersect + the OptiX programs
   (scene_optix.cpp:81-126, cuda/psdr_cuda.cu:9-45): closest hit with
   t in [1e-3, tmax] for m rays given as 7 SoA streams; writes mesh id,
   GLOBAL triangle id (-1 on miss) and the barycentrics of vertex 1 and 2. */
int psdr_trace(psdr_scene_t h, int32_t m,
               const float *ox, const float *oy, const float *oz,
               const float *dx, const float *dy, const float *dz,
               const float *tmax,
               int32_t *out_shape, int32_t *out_tri, float *out_u, float *out_v,
               void *stream);

/* Replaces Integrator::renderC (integrator.cpp:13-29): out_img [H*W*3],
   overwritten with (1/spp) * sum over this shard's slots. */
int psdr_render_c(psdr_scene_t h, const psdr_render_opts *opts,
                  float *out_img, void *stream);

/* Replaces Integrator::renderD followed by enoki forward(P)
   (integrator.cpp:32-60, examples/run_test.py:126-129): interior +
   primary-edge + secondary-edge terms; out_dimg [K][H*W*3]. */
int psdr_render_d_fwd(psdr_scene_t h, const psdr_render_opts *opts,
                      int32_t K, const psdr_tangents *tangents,
                      float *out_img, float *out_dimg, void *stream);

/* Replaces Integrator::renderD followed by enoki backward(loss)
   (docs/inverse_diff_render.rst): adj_img [H*W*3] = dLoss/dImage; gradients
   are scatter-added into `grads`. out_img may be NULL. */
int psdr_render_d_rev(psdr_scene_t h, const psdr_render_opts *opts,
                      const float *adj_img, float *out_img,
                      const psdr_grads *grads, void *stream);

/* Replaces DirectIntegrator::preprocess_secondary_edges (direct.cpp:166-204):
   out_mass [reso0*reso1*reso2] = mean over nrounds of per-cell max-RGB. */
int psdr_guide_build(psdr_scene_t h, const psdr_render_opts *opts,
                     const int32_t reso[4], int32_t nrounds,
                     float *out_mass, void *stream);

/* ---- the differentiable table chain of Scene::configure as kernels (csrc/psdr_tables.hip) -------------------------------------
   All pointers are device pointers; vertices [V][3], faces [T][3] int32 (global vertex ids), rows [T][row_stride >= 22] in the
   TriangleInfo layout (PSDR_TRI_STRIDE words when written straight into tri_info), edges [E][5] int32 = v0, v1, face0, face1
   (-1: boundary), opposite vertex of face0 (global ids; Mesh::m_edge_indices, mesh.cpp:154-196).  The *_rev entry points ADD the
   adjoints into a_v / a_rows / a_w2s (the caller zeroes them). */
/* World positions (Mesh::configure, src/shape/mesh.cpp:226-232: transform_pos of include/psdr/core/transform.h:84-88 by the mesh's to_world):
   v_raw [V][3] object-space positions, vmesh [V] int32 = the mesh of every vertex, mats [M][16] row-major to_world matrices.
   The reverse entry point WRITES a_raw (adjoint of v_raw; the matrices carry no gradient on this path: the caller keeps the torch chain when they do). */
int psdr_geo_world_vertices_fwd(int32_t V, const float *v_raw, const int32_t *vmesh, const float *mats, float *v_world, void *stream);
int psdr_geo_world_vertices_rev(int32_t V, const float *v_raw, const int32_t *vmesh, const float *mats, const float *v_world, const float *a_world, float *a_raw,
                                void *stream);
/* process_mesh (src/shape/mesh.cpp:20-51): rows = p0 e1 e2 n0 n1 n2 face_normal face_area with area-weighted vertex normals;
   vsum [V][3] = scratch kept for the adjoint (the un-normalised vertex normals). */
int psdr_geo_tri_rows_fwd(int32_t V, int32_t T, const float *v, const int32_t *faces, float *vsum, float *rows, int32_t row_stride, void *stream);
int psdr_geo_tri_rows_rev(int32_t V, int32_t T, const float *v, const int32_t *faces, const float *vsum, const float *a_rows, int32_t row_stride,
                          float *a_vsum /* scratch [V][3] */, float *a_v, void *stream);
/* SecondaryEdgeInfo of EVERY candidate edge (Mesh::configure, mesh.cpp:251-270): info [E][16] = p0 e1 n0 n1 p2 is_boundary, and
   keep [E] = 1 where the coplanar filter of scene.cpp:219-244 keeps the edge (the caller compacts). */
int psdr_geo_sec_edges_fwd(int32_t E, const int32_t *edges, const float *v, const float *rows, int32_t row_stride, float *info, uint8_t *keep, void *stream);
int psdr_geo_sec_edges_rev(int32_t E, const int32_t *edges, const float *a_info, float *a_v, float *a_rows, int32_t row_stride, void *stream);
/* PrimaryEdgeInfo of every candidate edge for one sensor (PerspectiveCamera::configure, src/sensor/perspective.cpp:39-111):
   cam22 = world_to_sample (16, row-major), camera position (3), viewing direction (3); face_normals [E] = 1 where the edge's mesh uses
   face normals; rows8 [E][PSDR_PEDGE_STRIDE], z4 [E][4] (psdr_scene_desc::prim_edge_z), keep [E] = the silhouette test. */
int psdr_geo_prim_edges_fwd(int32_t E, const int32_t *edges, const uint8_t *face_normals, const float *v, const float *rows, int32_t row_stride,
                            const float *cam22, float *rows8, float *z4, uint8_t *keep, void *stream);
int psdr_geo_prim_edges_rev(int32_t E, const int32_t *edges, const float *v, const float *cam22, const float *a_rows8, float *a_v, float *a_w2s /* [16] */,
                            void *stream);

/* The kept edges of a candidate table as a table of the SAME capacity, with the count left on the device (scene.cpp:219-244 and
   perspective.cpp:96-111 compress the kept edges and build a DiscreteDistribution over their lengths, pmf.cpp:7-21, inside one Enoki trace;
   an eager host chain would read the number of kept edges back to size the result).  rows [E][S], keep [E]; the weight of a row is its
   word w0 (wn = 1) or the norm of words w0..w0+2 (wn = 3).  aux [E][aux_stride] 32-bit words of which the first A travel with the row
   (A = 0: none).  Outputs: rows_out [E][S] / aux_out [E][A] = the kept rows in their order, zero behind them; pos [E] = the new row of
   edge e or -1; pmf / cmf [E] = weight / sum and its running sum (cmf = 1 from the last kept row on, so a lower-bound search for u < 1
   never leaves the kept rows; the host passes sum = 1); header [2] = {number of kept rows as int bits, sum of the weights}.
   scratch: 4 * ceil(E / 1024) words.  _rev WRITES a_rows [E][S] (the adjoint of rows) from a_rows_out. */
int psdr_geo_compact_edges_fwd(int32_t E, const float *rows, int32_t S, const uint8_t *keep, int32_t w0, int32_t wn, const void *aux, int32_t aux_stride, int32_t A,
                               void *scratch, float *rows_out, void *aux_out, int32_t *pos, float *pmf, float *cmf, float *header, void *stream);
int psdr_geo_compact_edges_rev(int32_t E, int32_t S, const int32_t *pos, const float *a_rows_out, float *a_rows, void *stream);
/* Mesh areas and the emitter tables (scene.cpp:183-196, area.cpp:10-16, mesh.cpp:244-249) without a host round trip: face_offset [M + 1],
   mesh_emitter [M] (-1: none), emitter_i [Ne][PSDR_EMITTER_I_STRIDE], radiance [Ne][3], env_weight [Ne] (< 0: an area light; otherwise the
   sampling weight of the environment map).  Outputs: mesh_area [M]; emitter_f [Ne][PSDR_EMITTER_F_STRIDE]; emitter_pmf / emitter_cmf [Ne]
   NORMALISED (the host passes emitter_sum = 1); face_pmf / face_cmf = the face-area distributions of the area lights' meshes at
   emitter_i[.][3] (unnormalised, their sums in emitter_f[.][5]). */
int psdr_geo_emitter_tables(int32_t M, const float *rows, int32_t row_stride, const int32_t *face_offset, const int32_t *mesh_emitter, int32_t Ne, const int32_t *emitter_i,
                            const float *radiance, const float *env_weight, float *mesh_area, float *emitter_f, float *emitter_pmf, float *emitter_cmf, float *face_pmf,
                            float *face_cmf, void *stream);

/* Counters of the last render call on this handle (host values):
   [0] rays traced, [1] camera slots, [2] primary-edge slots, [3] secondary-edge slots. */
int psdr_get_counters(psdr_scene_t h, uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* PSDR_HIP_H */
