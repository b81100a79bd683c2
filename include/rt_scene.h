/*
 * rt_scene.h -- flat, plain-old-data description of a parsed scene.
 *
 * This is what a `flatten()` of the reference's `Scene`
 * (raytracer-lib/src/scene/mod.rs:14-25) hands across the C boundary: every
 * value the render path reads, with the trait objects replaced by tagged
 * structs.  Geometry is f64 and colour is f32, exactly as the reference CLI
 * builds it (raytracer-cli/Cargo.toml:7, raytracer-lib/src/prelude.rs:10-19).
 *
 * Derived data (bounding boxes, centroids, triangle face normals, the BVHs)
 * is NOT carried here: each consumer rebuilds it from these fields with the
 * reference's constructors, so that two consumers cannot silently share one
 * wrong tree.
 *
 * Plain C, no pointers to anything but arrays of the structs below.
 */
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RtVec3 {
    double x, y, z;
} RtVec3;

/* camera/mod.rs:25-34 (CameraBase), camera/perspective.rs:5-8,
 * camera/orthographic.rs:6-8, math/coordinate_system.rs:4-9 */
enum { RT_CAMERA_PERSPECTIVE = 0, RT_CAMERA_ORTHOGRAPHIC = 1 };
typedef struct RtCamera {
    int32_t type;
    uint32_t pixels_x, pixels_y; /* camera/mod.rs:28-29 */
    uint32_t _pad;
    RtVec3 position;             /* basis.position */
    RtVec3 u, v, w;              /* basis.u / v / w  */
    double left, right, top, bottom;
    double focal_length;         /* perspective only */
} RtCamera;

/* shader/{lambertian,blinn_phong,perfect_mirror,normal,null,ggx_mirror}.rs */
enum {
    RT_SHADER_LAMBERTIAN = 0,
    RT_SHADER_BLINN_PHONG = 1,
    RT_SHADER_PERFECT_MIRROR = 2,
    RT_SHADER_NORMAL = 3,
    RT_SHADER_NULL = 4,
    RT_SHADER_GGX_MIRROR = 5 /* shader/ggx_mirror.rs: roughness (clamped to [0, 1]) and samples */
};
typedef struct RtShader {
    int32_t type;
    float diffuse[3];
    float specular[3];
    float shininess;
    double roughness;  /* GGX only */
    uint32_t samples;  /* GGX only */
    uint32_t _pad;
} RtShader;

/* light/{point,ambient}.rs */
enum { RT_LIGHT_POINT = 0, RT_LIGHT_AMBIENT = 1 };
typedef struct RtLight {
    int32_t type;
    float intensity[3];
    RtVec3 position; /* point only */
} RtLight;

/* geometry/mod.rs:19-26 */
enum {
    RT_SHAPE_SPHERE = 0,
    RT_SHAPE_BOX = 1,
    RT_SHAPE_TRIANGLE = 2,
    RT_SHAPE_MESH = 3,
    RT_SHAPE_INSTANCE = 4
};
typedef struct RtShape {
    int32_t type;
    int32_t shader;  /* index into RtScene.shaders */
    int32_t mesh;    /* MESH: index into RtScene.meshes, else -1 */
    int32_t proto;   /* INSTANCE: index into RtScene.protos, else -1 */
    /* SPHERE: p0 = center, radius.  BOX: p0 = min, p1 = max.
     * TRIANGLE: p0, p1, p2 = a, b, c. */
    RtVec3 p0, p1, p2;
    double radius;
    /* INSTANCE only (geometry/instance.rs:11-18), row-major 4x4. */
    double inv_transform[16];
    double normal_matrix[16];
    RtVec3 bbox_min, bbox_max; /* world-space box of the instance */
} RtShape;

/* geometry/mesh.rs:39-58: one triangle per index triple, in file order;
 * 9 doubles per triangle (a.xyz, b.xyz, c.xyz), positions already widened
 * from the OBJ loader's f32. */
typedef struct RtMesh {
    uint64_t n_triangles;
    const double *vertices;
} RtMesh;

typedef struct RtScene {
    RtCamera camera;
    float background[3];       /* scene/mod.rs:686-697 */
    uint16_t recursion_depth;  /* scene/mod.rs:710 */
    uint8_t disable_shadows;   /* scene/mod.rs:703 */
    uint8_t _pad0;
    uint32_t image_width, image_height;
    uint32_t n_shapes, n_protos, n_meshes, n_shaders, n_lights;
    uint32_t _pad1;
    const RtShape *shapes;   /* top-level shapes, scene order (scene/mod.rs:555-667) */
    const RtShape *protos;   /* instance prototypes (scene/mod.rs:497-546) */
    const RtMesh *meshes;
    const RtShader *shaders;
    const RtLight *lights;   /* scene order (scene/mod.rs:670-683) */
} RtScene;

#ifdef __cplusplus
}
#endif
#endif /* RT_SCENE_H */
