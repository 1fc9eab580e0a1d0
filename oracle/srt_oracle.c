/*
 * srt_oracle.c — CPU restatement of the reference's per-pixel trace/shade/accumulate
 * loop.  TEST INFRASTRUCTURE ONLY (see srt_oracle.h): never linked into, loaded by or
 * called from the product path.
 *
 * PARITY UNPINNED by reference output (the reference has no tests/fixtures and cannot
 * be built here).  Every function cites the reference lines it restates
 * (paths relative to /root/reference/Raytracer/).  The arithmetic follows the
 * reference operation by operation: same association, same int/float/double
 * promotions, Color's clamp-negatives constructor on every Color result.
 *
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math; contraction would change bits).
 * The reference is built by MSVC x64 /fp:precise without /arch:AVX2 (Raytracer.vcxproj:
 * 184-204): binary32 arithmetic, no FMA, FLT_EVAL_METHOD 0 — what gcc does on x86-64
 * with contraction off.
 */
#include "srt_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "srt_defs.h"

/* ------------------------------------------------------------------------------------
 * float3 (Common.hpp:22-179) — only the members the hot path uses.
 * ---------------------------------------------------------------------------------- */
typedef struct f3 {
    float x, y, z;
} f3;

static f3 f3_make(float x, float y, float z) { /* Common.hpp:72-76 */
    f3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
static f3 f3_splat(float x) { return f3_make(x, x, x); }                             /* :107-111 */
static f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }    /* :112-114 */
static f3 f3_neg(f3 a) { return f3_make(a.x * -1, a.y * -1, a.z * -1); }             /* :115-117 */
static f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }    /* :118-120 */
static f3 f3_mul(f3 a, f3 o) { return f3_make(a.x * o.x, a.y * o.y, a.z * o.z); }    /* :121-123 */
static f3 f3_div(f3 a, f3 o) { return f3_make(a.x / o.x, a.y / o.y, a.z / o.z); }    /* :124-126 */
/* `v * s` with a float s goes through the implicit float3(float) constructor */
static f3 f3_muls(f3 a, float s) { return f3_mul(a, f3_splat(s)); }
static float f3_magnitude(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); } /* :77-79 */
static float f3_magnitude_sqrd(f3 a) { return (a.x * a.x + a.y * a.y + a.z * a.z); } /* :80-82 */
static float f3_dot(f3 l, f3 r) { return (l.x * r.x + l.y * r.y + l.z * r.z); }      /* :83-93 */
static float flerpf(float a, float b, float t) { return a * (1 - t) + b * t; }       /* :19-21 */
static f3 f3_lerp(f3 a, f3 b, float t) {                                             /* :97-99 */
    return f3_make(flerpf(a.x, b.x, t), flerpf(a.y, b.y, t), flerpf(a.z, b.z, t));
}
static f3 f3_normalized(f3 a) { /* :159-162 */
    float length = f3_magnitude(a);
    return f3_make(a.x / length, a.y / length, a.z / length);
}
static f3 f3_reflect(f3 v, f3 normal) { /* :163-165 */
    return f3_sub(v, f3_muls(normal, 2 * f3_dot(v, normal)));
}
static f3 f3_max(f3 a, f3 o) { /* :167-172 */
    return f3_make(a.x > o.x ? a.x : o.x, a.y > o.y ? a.y : o.y, a.z > o.z ? a.z : o.z);
}
static f3 f3_yzx(f3 a) { return f3_make(a.y, a.z, a.x); } /* :65-67 */
static f3 f3_zxy(f3 a) { return f3_make(a.z, a.x, a.y); } /* :68-70 */

/* free functions, Common.hpp:328-351 */
static f3 f3_sign(f3 t) { /* :328-333 */
    return f3_make(t.x != 0 ? t.x / fabsf(t.x) : 0, t.y != 0 ? t.y / fabsf(t.y) : 0,
                   t.z != 0 ? t.z / fabsf(t.z) : 0);
}
static f3 f3_abs(f3 t) { return f3_make(fabsf(t.x), fabsf(t.y), fabsf(t.z)); } /* :334-336 */
static f3 f3_step(f3 edge, f3 t0) {                                            /* :337-339 */
    return f3_make(edge.x <= t0.x ? 1 : 0, edge.y <= t0.y ? 1 : 0, edge.z <= t0.z ? 1 : 0);
}
static float maxf_t(float a, float b) { return a > b ? a : b; } /* template max :344-347 */
static float minf_t(float a, float b) { return a < b ? a : b; } /* template min :348-351 */

/* ------------------------------------------------------------------------------------
 * Color (Common.hpp:180-280).  EVERY arithmetic result passes through the 4-argument
 * constructor, which clamps negatives to 0 (:253-262).  NaN and -0 pass unchanged.
 * ---------------------------------------------------------------------------------- */
typedef struct col {
    float r, g, b, a;
} col;

static col col_make(float r, float g, float b, float a) { /* :253-262 */
    col c;
    if (r < 0) r = 0;
    if (g < 0) g = 0;
    if (b < 0) b = 0;
    if (a < 0) a = 0;
    c.r = r;
    c.g = g;
    c.b = b;
    c.a = a;
    return c;
}
static col col_rgb(const float v[3]) { return col_make(v[0], v[1], v[2], 0); }
static col col_muls(col c, float s) { return col_make(c.r * s, c.g * s, c.b * s, c.a * s); } /* :209-211 */
static col col_mul(col c, col o) { return col_make(c.r * o.r, c.g * o.g, c.b * o.b, c.a * o.a); } /* :212-214 */
static col col_div(col c, col o) { return col_make(c.r / o.r, c.g / o.g, c.b / o.b, c.a / o.a); } /* :215-217 */
static col col_add(col c, col o) { return col_make(c.r + o.r, c.g + o.g, c.b + o.b, c.a + o.a); } /* :218-220 */
static col col_lerp(col a, col b, float time) { /* :275-279 (the two ?: statements are no-ops) */
    return col_make(a.r * (1 - time) + b.r * time, a.g * (1 - time) + b.g * time,
                    a.b * (1 - time) + b.b * time, a.a * (1 - time) + b.a * time);
}

/* (int)f as x86 cvttss2si does it (the reference is x64 MSVC): NaN and out-of-range
 * give 0x80000000.  Plain C would be undefined behaviour there. */
static int cvtt_x86(float f) {
    if (f != f || f >= 2147483648.0f || f < -2147483648.0f) return INT_MIN;
    return (int)f;
}

/* Color::fromRGBA / operator Uint32 (Common.hpp:189-208) */
static uint32_t col_pack(col c) {
    int sr = cvtt_x86(c.r * 255);
    int sg = cvtt_x86(c.g * 255);
    int sb = cvtt_x86(c.b * 255);
    int sa = cvtt_x86(c.a * 255);
    if (sr > 255) sr = 255;
    if (sg > 255) sg = 255;
    if (sb > 255) sb = 255;
    if (sa > 255) sa = 255;
    uint8_t sr8 = (uint8_t)sr, sg8 = (uint8_t)sg, sb8 = (uint8_t)sb, sa8 = (uint8_t)sa;
    return (uint32_t)((uint32_t)sa8 << 24 | (uint32_t)sr8 << 16 | (uint32_t)sg8 << 8 | (uint32_t)sb8);
}

/* ------------------------------------------------------------------------------------
 * Rayhit (Common.hpp:320-325)
 * ---------------------------------------------------------------------------------- */
typedef struct rayhit {
    int valid;
    f3 normal;
    f3 point;
    float distance;
} rayhit;

static rayhit rayhit_default(void) {
    rayhit h;
    h.valid = 0;
    h.normal = f3_make(0, 0, 0);
    h.point = f3_make(0, 0, 0);
    h.distance = 0; /* uninitialised in the reference; never read while !valid */
    return h;
}

/* Sphere::line_sphere_intersection + Sphere::Raytrace (Object.hpp:104-141, 153-167) */
static rayhit sphere_raytrace(const srt_object* o, f3 rayPosition, f3 rayDirection) {
    rayhit hitResults = rayhit_default();
    f3 spherePosition = f3_make(o->position[0], o->position[1], o->position[2]);
    float sphereRadius = o->radius;

    f3 directionToSphere = f3_sub(spherePosition, rayPosition);          /* :115 */
    /* :116 computes |directionToSphere| and never uses it */
    float tc = f3_dot(directionToSphere, rayDirection);                  /* :118 */
    tc = fabsf(tc);                                                      /* :119 */
    f3 intersectCheck = f3_add(f3_muls(rayDirection, tc), rayPosition);  /* :121 */
    float squaredRadius = sphereRadius * sphereRadius;                   /* :122 */
    f3 dir2sphereIntersectDir = f3_sub(intersectCheck, spherePosition);  /* :124 */
    float d2 = f3_magnitude_sqrd(dir2sphereIntersectDir);                /* :125 */
    if (d2 > squaredRadius) return hitResults;                           /* :127-129 */
    float t1c = sqrtf(squaredRadius - d2);                               /* :131 */
    float t1 = tc - t1c;                                                 /* :133 */
    f3 outPosition = f3_add(rayPosition, f3_muls(rayDirection, t1));     /* :136 */
    f3 outNormal = f3_normalized(f3_sub(outPosition, spherePosition));   /* :137 */
    hitResults.normal = outNormal;
    hitResults.point = outPosition;
    hitResults.distance = t1;
    hitResults.valid = 1;
    return hitResults;
}

/* Box::iBox (Object.hpp:173-200) */
static float ibox(f3 ro, f3 rd, f3 distBound, f3* normal, f3 boxSize) {
    f3 m = f3_div(f3_sign(rd), f3_max(f3_abs(rd), f3_splat((float)1e-8))); /* :175 */
    f3 n = f3_mul(m, ro);                                                  /* :176 */
    f3 k = f3_mul(f3_abs(m), boxSize);                                     /* :177 */
    f3 t1 = f3_sub(f3_neg(n), k);                                          /* :179 */
    f3 t2 = f3_add(f3_neg(n), k);                                          /* :180 */
    float tN = maxf_t(maxf_t(t1.x, t1.y), t1.z);                           /* :181 */
    float tF = minf_t(minf_t(t2.x, t2.y), t2.z);                           /* :182 */
    if (tN > tF || tF <= 0.) {                                             /* :184 */
        return FLT_MAX;
    } else {
        if (tN >= distBound.x && tN <= distBound.y) {                      /* :188-191 */
            *normal = f3_mul(f3_mul(f3_neg(f3_sign(rd)), f3_step(f3_yzx(t1), t1)),
                             f3_step(f3_zxy(t1), t1));
            return tN;
        } else if (tF >= distBound.x && tF <= distBound.y) {               /* :192-195 */
            *normal = f3_mul(f3_mul(f3_neg(f3_sign(rd)), f3_step(f3_yzx(t1), t1)),
                             f3_step(f3_zxy(t1), t1));
            return tF;
        } else {
            return FLT_MAX;
        }
    }
}

/* Box::Raytrace (Object.hpp:224-233) */
static rayhit box_raytrace(const srt_object* o, f3 rayOrigin, f3 rayDir) {
    f3 normal = f3_make(0, 0, 0);
    f3 position = f3_make(o->position[0], o->position[1], o->position[2]);
    f3 size = f3_make(o->half_size[0], o->half_size[1], o->half_size[2]);
    float dist = ibox(f3_sub(rayOrigin, position), rayDir, f3_make((float)0.01, 10000, 0), &normal, size);
    rayhit ret = rayhit_default();
    ret.normal = normal;
    ret.point = f3_add(rayOrigin, f3_muls(rayDir, dist));
    ret.distance = dist;
    ret.valid = dist == FLT_MAX ? 0 : 1;
    return ret;
}

/* ------------------------------------------------------------------------------------
 * EXTENSION — triangle meshes.  NOT a restatement: the reference has no triangle primitive
 * (Object.hpp defines Sphere and Box only).  This is the project's own definition
 * (include/srt_pathtrace.h), written in the reference's style: binary32, no FMA,
 * fixed association, float3 value semantics.
 * ---------------------------------------------------------------------------------- */
static f3 f3_cross(f3 a, f3 b) {
    return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* Moller-Trumbore for the triangle (v0, v0+e1, v0+e2) */
static rayhit triangle_raytrace(f3 v0, f3 e1, f3 e2, f3 rayOrigin, f3 rayDir) {
    rayhit h = rayhit_default();
    f3 pvec = f3_cross(rayDir, e2);
    float det = f3_dot(e1, pvec);
    if (fabsf(det) < 1e-12f) return h; /* ray parallel to the plane (also false for NaN) */
    if (!(fabsf(det) >= 1e-12f)) return h;
    float inv = 1.0f / det;
    f3 tvec = f3_sub(rayOrigin, v0);
    float u = f3_dot(tvec, pvec) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return h;
    f3 qvec = f3_cross(tvec, e1);
    float v = f3_dot(rayDir, qvec) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return h;
    float t = f3_dot(e2, qvec) * inv;
    if (!(t >= (float)0.01 && t <= 10000.0f)) return h; /* the Box distance bounds, Object.hpp:226 */
    f3 n = f3_normalized(f3_cross(e1, e2));
    if (f3_dot(n, rayDir) > 0) n = f3_muls(n, -1); /* face the ray */
    h.valid = 1;
    h.distance = t;
    h.point = f3_add(rayOrigin, f3_muls(rayDir, t));
    h.normal = n;
    return h;
}

/* a mesh object: closest valid triangle; strict '<' keeps the lower triangle index on ties */
static rayhit mesh_raytrace(const srt_object* o, const srt_mesh* meshes, size_t mesh_count, f3 origin, f3 dir) {
    rayhit best = rayhit_default();
    if (!meshes || o->mesh < 0 || (size_t)o->mesh >= mesh_count) return best;
    const srt_mesh* m = &meshes[o->mesh];
    f3 pos = f3_make(o->position[0], o->position[1], o->position[2]);
    float shortest = INFINITY;
    for (size_t k = 0; k < m->triangle_count; ++k) {
        const uint32_t* ix = m->indices + 3 * k;
        if (ix[0] >= m->vertex_count || ix[1] >= m->vertex_count || ix[2] >= m->vertex_count) continue;
        const float *a = m->vertices + 3 * (size_t)ix[0], *b = m->vertices + 3 * (size_t)ix[1], *c = m->vertices + 3 * (size_t)ix[2];
        f3 v0 = f3_add(f3_make(a[0], a[1], a[2]), pos); /* world vertex = vertex + position */
        f3 v1 = f3_add(f3_make(b[0], b[1], b[2]), pos);
        f3 v2 = f3_add(f3_make(c[0], c[1], c[2]), pos);
        rayhit h = triangle_raytrace(v0, f3_sub(v1, v0), f3_sub(v2, v0), origin, dir);
        if (h.valid && h.distance < shortest) {
            shortest = h.distance;
            best = h;
        }
    }
    return best;
}

/* virtual Object::Raytrace dispatch (Object.hpp:21-23,153,224) */
static rayhit object_raytrace_m(const srt_object* o, const srt_mesh* meshes, size_t mesh_count, f3 origin, f3 dir) {
    if (o->type == SRT_OBJ_SPHERE) return sphere_raytrace(o, origin, dir);
    if (o->type == SRT_OBJ_BOX) return box_raytrace(o, origin, dir);
    if (o->type == SRT_OBJ_MESH) return mesh_raytrace(o, meshes, mesh_count, origin, dir);
    return rayhit_default();
}
static rayhit object_raytrace(const srt_object* o, f3 origin, f3 dir) { return object_raytrace_m(o, NULL, 0, origin, dir); }

/* ------------------------------------------------------------------------------------
 * Render context: the globals of Raytracer.cpp:30-35,46-48,55-61
 * ---------------------------------------------------------------------------------- */
typedef struct octx {
    const srt_object* objects;
    size_t count;
    const srt_mesh* meshes;
    size_t mesh_count;
    const srt_environment* env;
    int32_t width, height; /* SCREEN_WIDTH / SCREEN_HEIGHT :26-27 */
    int32_t fov;           /* FOV :31 */
    int32_t max_bounces;   /* MAXBOUNCES :32 */
    int32_t pow_mode;
    int32_t simpledraw;      /* SIMPLEDRAW :35 */
    int32_t selected_object; /* selectedObject :53 as a list index, -1 = NULL */
    /* rand() replacement: see include/srt_defs.h */
    uint32_t rng_key, rng_draws;
    uint64_t rays;
} octx;

static int o_rand(octx* c) { return (int)srt_rng_draw(c->rng_key, c->rng_draws++); }

static float o_powf(const octx* c, float x, float y) {
    return c->pow_mode == SRT_ORACLE_POW_SHARED ? srt_powf(x, y) : powf(x, y);
}

typedef struct rayhitobject { /* Object.hpp:81-84 */
    rayhit rayHit;
    int objectReference; /* index; uninitialised pointer on a miss in the reference */
} rayhitobject;

/* GetClosestObject (Raytracer.cpp:123-140) */
static rayhitobject get_closest_object(octx* c, f3 rayOrigin, f3 rayDirection) {
    rayhitobject returnResults;
    returnResults.rayHit = rayhit_default();
    returnResults.objectReference = -1;
    float shortestDistance = INFINITY; /* :126 */
    c->rays++;
    for (size_t i = 0; i < c->count; i++) { /* :127 */
        rayhit hitResults = object_raytrace_m(&c->objects[i], c->meshes, c->mesh_count, rayOrigin, rayDirection);
        if (hitResults.valid) {                            /* :130 */
            if (hitResults.distance < shortestDistance) {  /* :132 strict: first wins ties */
                returnResults.objectReference = (int)i;
                shortestDistance = hitResults.distance;
                returnResults.rayHit = hitResults;
            }
        }
    }
    return returnResults;
}

/* GetEnvironmentColor (Raytracer.cpp:77-89) with the globals of :55-59 */
static col get_environment_color(const octx* c, f3 rayDirection) {
    const srt_environment* e = c->env;
    f3 SunDirection = f3_make(e->sun_direction[0], e->sun_direction[1], e->sun_direction[2]);
    col SkyColor = col_rgb(e->sky_color), HorizonColor = col_rgb(e->horizon_color);
    col GroundColor = col_rgb(e->ground_color), SunColor = col_rgb(e->sun_color);

    float upd = f3_dot(rayDirection, f3_make(0, 1, 0));                                      /* :78 */
    /* :79 — `> 0.99` compares the float dot product with a DOUBLE literal */
    col Sun = (double)f3_dot(rayDirection, f3_muls(SunDirection, -1)) > 0.99 ? SunColor
                                                                             : col_make(0, 0, 0, 0);
    if (upd > 0) {
        col t = col_lerp(HorizonColor, SkyColor, o_powf(c, upd, 0.1f));                      /* :81 */
        t = col_lerp(t, col_muls(SkyColor, 0.1f), upd);                                      /* :82 */
        return col_add(t, Sun);                                                              /* :83 */
    } else {
        upd = fabsf(upd);                                                                    /* :86 */
        return col_add(col_lerp(HorizonColor, GroundColor, o_powf(c, upd, .05f)), Sun);      /* :87 */
    }
}

/* GetRandomDirection (Raytracer.cpp:90-98): |component| <= 1 so the do/while never repeats */
static f3 get_random_direction(octx* c) {
    f3 sr;
    do {
        sr.x = ((float)o_rand(c) / SRT_RAND_MAX - 0.5f) * 2;
        sr.y = ((float)o_rand(c) / SRT_RAND_MAX - 0.5f) * 2;
        sr.z = ((float)o_rand(c) / SRT_RAND_MAX - 0.5f) * 2;
    } while (sr.x * sr.y * sr.z > 1);
    return f3_normalized(sr);
}

/* GetRandomNormalOrientedHemisphere (Raytracer.cpp:99-105) */
static f3 get_random_normal_oriented_hemisphere(octx* c, f3 normal) {
    f3 sr = get_random_direction(c);
    if (f3_dot(sr, normal) < 0) {
        sr = f3_muls(sr, -1);
    }
    return sr;
}

/* GetRayDirection (Raytracer.cpp:106-122) */
static f3 get_ray_direction(const srt_camera* cam, int32_t W, int32_t H, int32_t fov, int pixelX,
                            int pixelY) {
    const float clipDistance = .01f;
    f3 right = f3_make(cam->right[0], cam->right[1], cam->right[2]);
    f3 up = f3_make(cam->up[0], cam->up[1], cam->up[2]);
    f3 forward = f3_make(cam->forward[0], cam->forward[1], cam->forward[2]);

    float nX = (pixelX / (float)W) * 2 - 1;                      /* :109 */
    float nY = (pixelY / (float)H) * 2 - 1;                      /* :110 */
    float aspecRatio = (float)W / (float)H;                      /* :111 */
    float hFov = (float)(fov * 3.14159265358979323846 / 180.0f); /* :112 int*double/float -> double, stored float */
    f3 forwardDir = f3_muls(forward, clipDistance);              /* :113 */
    float rd = (clipDistance * tanf(hFov / 2.0f)) * aspecRatio;  /* :114 */
    float ld = (clipDistance * tanf(hFov / 2.0f));               /* :115 */
    f3 u = f3_muls(f3_muls(right, rd), nX);                      /* :116 */
    f3 v = f3_muls(f3_muls(up, ld), nY);                         /* :117 */
    return f3_normalized(f3_add(f3_add(u, v), forwardDir));      /* :119 */
}

static const srt_material* mat_of(const octx* c, int idx) { return &c->objects[idx].material; }

/* template smoothstep (Common.hpp:352-365) */
static float smoothstep_t(float edge0, float edge1, float x) {
    if (x < edge0) return 0;
    if (x >= edge1) return 1;
    x = (x - edge0) / (edge1 - edge0);
    return x * x * (3 - 2 * x);
}

/* RaytraceScene, SIMPLEDRAW == false (Raytracer.cpp:141-146, 162-185, 212) */
static col raytrace_scene(octx* c, f3 rayOrigin, f3 rayDirection) {
    rayhitobject hit = get_closest_object(c, rayOrigin, rayDirection); /* :142 */
    if (!hit.rayHit.valid) {
        return get_environment_color(c, rayDirection);                 /* :143-145 */
    }
    if (c->simpledraw) { /* :147-160 */
        col reflectedColor = get_environment_color(c, f3_reflect(rayDirection, hit.rayHit.normal)); /* :148 */
        float k = mat_of(c, hit.objectReference)->specular_amount;                                 /* :149 */
        float s = mat_of(c, hit.objectReference)->smoothness;                                      /* :150 */
        float fresnal = 0;                                                                         /* :152 */
        if (hit.objectReference == c->selected_object) {                                           /* :153 */
            fresnal = 1 - f3_dot(f3_neg(hit.rayHit.normal), rayDirection);                         /* :154 */
            fresnal = maxf_t(fresnal, 0.0f);                                                       /* :155 */
            fresnal = smoothstep_t(0.0f, 0.5f, fresnal);                                           /* :156 */
        }
        col base = col_rgb(mat_of(c, hit.objectReference)->base_color);
        col emis = col_rgb(mat_of(c, hit.objectReference)->emissive_color);
        return col_lerp(col_add(col_add(col_muls(base, (1 - k)), col_muls(col_muls(reflectedColor, k), s)), emis),
                        col_make(3, 3, 0, 0), fresnal);                                            /* :159 */
    }
    col incomingLight = col_rgb(mat_of(c, hit.objectReference)->emissive_color); /* :162 */
    col hitColor = col_rgb(mat_of(c, hit.objectReference)->base_color);          /* :163 */
    f3 sray = rayDirection;                                                      /* :164 */
    int specularProb =
        mat_of(c, hit.objectReference)->specular_amount >= ((float)o_rand(c) / SRT_RAND_MAX); /* :165 */
    const float lightEnergyDissipation = 0.8f;                                                /* :166 */
    for (int i = 0; i < c->max_bounces; i++) {                                                /* :167 */
        if (i != 0) {
            hitColor = col_muls(hitColor, lightEnergyDissipation);                            /* :169-171 */
        }
        f3 reflectedRay = f3_reflect(sray, hit.rayHit.normal);                                /* :172 */
        sray = get_random_normal_oriented_hemisphere(c, hit.rayHit.normal);                   /* :174 */
        sray = f3_lerp(sray, reflectedRay,
                       mat_of(c, hit.objectReference)->smoothness * specularProb);            /* :175 */
        sray = f3_normalized(sray);                                                           /* :176 */
        hit = get_closest_object(
            c, f3_add(hit.rayHit.point, f3_muls(hit.rayHit.normal, .00001f)), sray);          /* :177 */
        if (!hit.rayHit.valid) {
            incomingLight = col_add(incomingLight, col_mul(get_environment_color(c, sray), hitColor)); /* :179 */
            break;
        }
        specularProb =
            mat_of(c, hit.objectReference)->specular_amount >= ((float)o_rand(c) / SRT_RAND_MAX); /* :182 */
        incomingLight = col_add(
            incomingLight, col_mul(col_rgb(mat_of(c, hit.objectReference)->emissive_color), hitColor)); /* :183 */
        hitColor = col_mul(hitColor, col_lerp(col_rgb(mat_of(c, hit.objectReference)->base_color),
                                              col_rgb(mat_of(c, hit.objectReference)->specular_color),
                                              specularProb));                                 /* :184 */
    }
    return incomingLight; /* :212 */
}

/* SetScreenPixel (Raytracer.cpp:63-76): accumulate half */
static col accumulate(col colorBuffer, col color, int setFrame, uint32_t ACCUMULATIONFRAMES) {
    if (!setFrame) {
        float weight = (float)(1.0 / (int)ACCUMULATIONFRAMES);                       /* :66 double divide */
        return col_add(col_muls(colorBuffer, (1 - weight)), col_muls(color, weight)); /* :67 */
    }
    return color; /* :70 */
}
/* SetScreenPixel: tone-map + pack half (:73-75) */
static uint32_t tonemap_pack(col finalColor) {
    finalColor = col_div(finalColor, col_add(col_make(1, 1, 1, 0), finalColor)); /* :74 */
    return col_pack(finalColor);                                                  /* :75 */
}

/* ------------------------------------------------------------------------------------
 * renderArea (Raytracer.cpp:223-257) for `sample_count` successive frames.  Pixels are
 * independent, so looping frames inside the pixel equals the reference's frame loop.
 * ---------------------------------------------------------------------------------- */
typedef struct worker {
    srt_oracle_job* job;
    int32_t minX, maxX, minY, maxY; /* scene coordinates, max exclusive */
    int x_outer;                    /* reference walk order :235-237 */
    int y_step;                     /* row interleave: thread t takes rows minY, minY + y_step, ... */
    int literal_blocks;             /* run renderArea's own block loop nest (SPLIT_REF_COLS + steps) */
    int pix_first, pix_step;        /* pix_step > 0: pixels of the window dealt round-robin to the threads */
    uint64_t rays;
} worker;

/* Anchor of the steps x steps block that pixel (x, y) belongs to in renderArea's walk
 * (:235-248): blocks start at the worker's minX (= stripe start) and at minY = 0. */
static void block_anchor(const srt_render_params* p, int x, int y, int* ax, int* ay) {
    int steps = p->steps > 1 ? p->steps : 1;
    int s0 = p->stripe_width > 0 ? (x / p->stripe_width) * p->stripe_width : 0;
    *ax = s0 + ((x - s0) / steps) * steps;
    *ay = (y / steps) * steps;
}

static void render_pixel(octx* c, const srt_oracle_job* job, int x, int y) {
    const srt_render_params* p = &job->params;
    const srt_camera* cam = job->camera;
    const int W = job->width, H = job->height;
    float* cb = job->accumulator + ((size_t)x + (size_t)y * W) * 4;
    col colorBuffer = {cb[0], cb[1], cb[2], cb[3]};
    f3 origin = f3_make(cam->position[0], cam->position[1], cam->position[2]);
    for (uint32_t s = 0; s < p->sample_count; s++) {
        uint32_t frame = p->first_sample + s;
        int ax, ay;
        block_anchor(p, x, y, &ax, &ay); /* the block's ray is traced through its anchor pixel (:239) */
        c->rng_key = srt_rng_key(p->seed, (uint32_t)(ax + ay * W), frame);
        c->rng_draws = 0;
        f3 rayDirection = get_ray_direction(cam, W, H, cam->fov_degrees, ax, ay); /* :239 */
        col color = raytrace_scene(c, origin, rayDirection);                     /* :240 */
        int setFrame = (s == 0) && (p->flags & SRT_RENDER_RESET);
        colorBuffer = accumulate(colorBuffer, color, setFrame, frame);           /* :246 -> :63 */
    }
    cb[0] = colorBuffer.r;
    cb[1] = colorBuffer.g;
    cb[2] = colorBuffer.b;
    cb[3] = colorBuffer.a;
    if (job->framebuffer) job->framebuffer[(size_t)(H - 1 - y) * W + x] = tonemap_pack(colorBuffer); /* :64,75 */
}

static void* worker_main(void* arg) {
    worker* w = (worker*)arg;
    srt_oracle_job* job = w->job;
    octx c;
    memset(&c, 0, sizeof c);
    c.objects = job->objects;
    c.count = job->object_count;
    c.meshes = job->meshes;
    c.mesh_count = job->mesh_count;
    c.env = job->env;
    c.width = job->width;
    c.height = job->height;
    c.fov = job->camera->fov_degrees;
    c.max_bounces = job->params.max_bounces;
    c.pow_mode = job->pow_mode;
    c.simpledraw = (job->params.flags & SRT_RENDER_PREVIEW) != 0;
    c.selected_object = job->params.selected_object;
    if (w->literal_blocks) { /* renderArea's own loop nest, :235-248, for ONE frame */
        const srt_render_params* p = &job->params;
        const int W = job->width, H = job->height, steps = p->steps > 1 ? p->steps : 1;
        const srt_camera* cam = job->camera;
        f3 origin = f3_make(cam->position[0], cam->position[1], cam->position[2]);
        for (int i = w->minX; i < w->maxX; i += steps) {
            for (int j = w->minY; j < w->maxY; j += steps) {
                c.rng_key = srt_rng_key(p->seed, (uint32_t)(i + j * W), p->first_sample);
                c.rng_draws = 0;
                col color = raytrace_scene(&c, origin, get_ray_direction(cam, W, H, cam->fov_degrees, i, j));
                for (int i1 = 0; i1 < steps && i + i1 < w->maxX; i1++) {
                    for (int j1 = 0; j1 < steps && j + j1 < w->maxY; j1++) {
                        float* cb = job->accumulator + ((size_t)(i + i1) + (size_t)(j + j1) * W) * 4;
                        col buf = {cb[0], cb[1], cb[2], cb[3]};
                        buf = accumulate(buf, color, (p->flags & SRT_RENDER_RESET) != 0, p->first_sample);
                        cb[0] = buf.r, cb[1] = buf.g, cb[2] = buf.b, cb[3] = buf.a;
                        if (job->framebuffer) job->framebuffer[(size_t)(H - 1 - (j + j1)) * W + (i + i1)] = tonemap_pack(buf);
                    }
                }
            }
        }
        w->rays = c.rays;
        return NULL;
    }
    if (w->pix_step > 0) { /* small window of a large frame: one pixel at a time, round-robin */
        const int ww = w->maxX - w->minX, n = ww * (w->maxY - w->minY);
        for (int i = w->pix_first; i < n; i += w->pix_step) render_pixel(&c, job, w->minX + i % ww, w->minY + i / ww);
    } else if (w->x_outer) {
        for (int x = w->minX; x < w->maxX; x++)
            for (int y = w->minY; y < w->maxY; y++) render_pixel(&c, job, x, y);
    } else {
        const int step = w->y_step > 0 ? w->y_step : 1;
        for (int y = w->minY; y < w->maxY; y += step)
            for (int x = w->minX; x < w->maxX; x++) render_pixel(&c, job, x, y);
    }
    w->rays = c.rays;
    return NULL;
}

static int imin(int a, int b) { return a < b ? a : b; }

int srt_oracle_render(srt_oracle_job* job) {
    if (!job || !job->env || !job->camera || !job->accumulator) return SRT_ERR_INVALID_ARG;
    if (job->object_count && !job->objects) return SRT_ERR_INVALID_ARG;
    const srt_render_params* p = &job->params;
    const int W = job->width, H = job->height;
    if (W <= 0 || H <= 0 || p->row_begin < 0 || p->row_end > H || p->row_begin > p->row_end)
        return SRT_ERR_INVALID_ARG;
    if (p->first_sample < 1 || p->sample_count < 1 || p->max_bounces < 0) return SRT_ERR_INVALID_ARG;
    int T = job->threads < 1 ? 1 : job->threads;
    /* memory rows [row_begin,row_end) = scene rows [H-row_end, H-row_begin) */
    const int y0 = H - p->row_end, y1 = H - p->row_begin;

    worker* ws = (worker*)calloc((size_t)T, sizeof(worker));
    pthread_t* th = (pthread_t*)calloc((size_t)T, sizeof(pthread_t));
    if (!ws || !th) {
        free(ws);
        free(th);
        return SRT_ERR_OOM;
    }
    int n = 0;
    /* test aid: a column window [col_begin, col_end) of the band (0,0 = all columns).  Only those pixels
     * are rendered — the frame, the camera and the RNG keys are the full frame's — so a few pixels of a
     * 4K / 1024-spp frame can be checked against the GPU without rendering the rest on the CPU. */
    const int window = job->col_end > job->col_begin;
    if (window && (job->col_begin < 0 || job->col_end > W)) {
        free(ws);
        free(th);
        return SRT_ERR_INVALID_ARG;
    }
    if (window) {
        for (int i = 0; i < T; i++) {
            ws[n].job = job;
            ws[n].minX = job->col_begin;
            ws[n].maxX = job->col_end;
            ws[n].minY = y0;
            ws[n].maxY = y1;
            ws[n].pix_first = i;
            ws[n].pix_step = T;
            n++;
        }
    } else if (job->split == SRT_ORACLE_SPLIT_REF_COLS) {
        int div = (int)ceil((double)(W / T)) + 1; /* Raytracer.cpp:330 — integer divide first */
        for (int i = 0; i < T; i++) {
            int initialX = div * i, nextX = imin(initialX + div, W); /* :338-340 */
            if (initialX >= nextX) continue;
            ws[n].job = job;
            ws[n].minX = initialX;
            ws[n].maxX = nextX;
            ws[n].minY = y0;
            ws[n].maxY = y1;
            ws[n].x_outer = 1;
            /* with block replication the literal loop nest is used (one frame, full height) */
            ws[n].literal_blocks = p->steps > 1 && p->sample_count == 1 && y0 == 0 && y1 == H;
            n++;
        }
    } else {
        /* rows are dealt round-robin to the threads: sky rows are ~10x cheaper than floor rows, so
         * contiguous bands would leave most threads idle (results do not depend on the split) */
        int rows = y1 - y0;
        for (int i = 0; i < T && i < rows; i++) {
            ws[n].job = job;
            ws[n].minX = 0;
            ws[n].maxX = W;
            ws[n].minY = y0 + i;
            ws[n].maxY = y1;
            ws[n].y_step = T;
            ws[n].x_outer = 0;
            n++;
        }
    }
    int started = 0;
    for (int i = 1; i < n; i++) {
        if (pthread_create(&th[i], NULL, worker_main, &ws[i]) != 0) break;
        started = i;
    }
    if (n > 0) worker_main(&ws[0]);
    for (int i = 1; i <= started; i++) pthread_join(th[i], NULL);
    for (int i = started + 1; i < n; i++) worker_main(&ws[i]); /* thread creation failed: run inline */
    uint64_t rays = 0;
    for (int i = 0; i < n; i++) rays += ws[i].rays;
    job->rays_out = rays;
    free(ws);
    free(th);
    return SRT_OK;
}

/* ------------------------------------------------------------------------------------
 * per-function probes
 * ---------------------------------------------------------------------------------- */
void srt_oracle_ray_direction(const srt_camera* cam, int32_t width, int32_t height, int32_t px,
                              int32_t py, float out_dir[3]) {
    f3 d = get_ray_direction(cam, width, height, cam->fov_degrees, px, py);
    out_dir[0] = d.x;
    out_dir[1] = d.y;
    out_dir[2] = d.z;
}

static void put3(float o[3], f3 v) {
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
}

int srt_oracle_intersect(const srt_object* obj, const float origin[3], const float dir[3],
                         float out_normal[3], float out_point[3], float* out_distance) {
    rayhit h = object_raytrace(obj, f3_make(origin[0], origin[1], origin[2]), f3_make(dir[0], dir[1], dir[2]));
    put3(out_normal, h.normal);
    put3(out_point, h.point);
    *out_distance = h.distance;
    return h.valid;
}

int srt_oracle_triangle(const float v0[3], const float v1[3], const float v2[3], const float origin[3],
                        const float dir[3], float out_normal[3], float out_point[3], float* out_distance) {
    f3 a = f3_make(v0[0], v0[1], v0[2]), b = f3_make(v1[0], v1[1], v1[2]), c = f3_make(v2[0], v2[1], v2[2]);
    rayhit h = triangle_raytrace(a, f3_sub(b, a), f3_sub(c, a), f3_make(origin[0], origin[1], origin[2]),
                                 f3_make(dir[0], dir[1], dir[2]));
    put3(out_normal, h.normal);
    put3(out_point, h.point);
    *out_distance = h.distance;
    return h.valid;
}

int srt_oracle_closest(const srt_object* objects, size_t count, const float origin[3],
                       const float dir[3], float out_normal[3], float out_point[3],
                       float* out_distance) {
    octx c;
    memset(&c, 0, sizeof c);
    c.objects = objects;
    c.count = count;
    rayhitobject h =
        get_closest_object(&c, f3_make(origin[0], origin[1], origin[2]), f3_make(dir[0], dir[1], dir[2]));
    put3(out_normal, h.rayHit.normal);
    put3(out_point, h.rayHit.point);
    *out_distance = h.rayHit.valid ? h.rayHit.distance : INFINITY;
    return h.rayHit.valid ? h.objectReference : -1;
}

void srt_oracle_environment(const srt_environment* env, const float dir[3], int32_t pow_mode,
                            float out_rgb[3]) {
    octx c;
    memset(&c, 0, sizeof c);
    c.env = env;
    c.pow_mode = pow_mode;
    col e = get_environment_color(&c, f3_make(dir[0], dir[1], dir[2]));
    out_rgb[0] = e.r;
    out_rgb[1] = e.g;
    out_rgb[2] = e.b;
}

void srt_oracle_trace_sample(const srt_object* objects, size_t count, const srt_environment* env,
                             const srt_camera* cam, int32_t width, int32_t height, int32_t px,
                             int32_t py, uint32_t sample, int32_t max_bounces, uint32_t seed,
                             int32_t pow_mode, float out_rgba[4], uint32_t* rays,
                             uint32_t* draws) {
    octx c;
    memset(&c, 0, sizeof c);
    c.objects = objects;
    c.count = count;
    c.env = env;
    c.width = width;
    c.height = height;
    c.fov = cam->fov_degrees;
    c.max_bounces = max_bounces;
    c.pow_mode = pow_mode;
    c.rng_key = srt_rng_key(seed, (uint32_t)(px + py * width), sample);
    f3 d = get_ray_direction(cam, width, height, cam->fov_degrees, px, py);
    col r = raytrace_scene(&c, f3_make(cam->position[0], cam->position[1], cam->position[2]), d);
    out_rgba[0] = r.r;
    out_rgba[1] = r.g;
    out_rgba[2] = r.b;
    out_rgba[3] = r.a;
    if (rays) *rays = (uint32_t)c.rays;
    if (draws) *draws = c.rng_draws;
}

uint32_t srt_oracle_tonemap_pack(const float rgba[4]) {
    col c = {rgba[0], rgba[1], rgba[2], rgba[3]};
    return tonemap_pack(c);
}

float srt_oracle_powf_shared(float x, float y) { return srt_powf(x, y); }

/* Raytracer.cpp:55-59 and the normalisation at :264 */
void srt_oracle_environment_default(srt_environment* env) {
    f3 SunDirection = f3_normalized(f3_make(1, -1, -1));               /* :55, :264 */
    col SkyColor = col_muls(col_make((float).2, (float).35, 1.0f, 0), 10.0f);    /* :56 */
    col HorizonColor = col_muls(col_make((float)1.0, 0.9f, 0.5f, 0), 5.0f);       /* :57 */
    col GroundColor = col_make(.08f, .06f, .03f, 0);                              /* :58 */
    col SunColor = col_make(500, 500, 500, 0);                                    /* :59 */
    put3(env->sun_direction, SunDirection);
    env->sky_color[0] = SkyColor.r, env->sky_color[1] = SkyColor.g, env->sky_color[2] = SkyColor.b;
    env->horizon_color[0] = HorizonColor.r, env->horizon_color[1] = HorizonColor.g,
    env->horizon_color[2] = HorizonColor.b;
    env->ground_color[0] = GroundColor.r, env->ground_color[1] = GroundColor.g,
    env->ground_color[2] = GroundColor.b;
    env->sun_color[0] = SunColor.r, env->sun_color[1] = SunColor.g, env->sun_color[2] = SunColor.b;
}
