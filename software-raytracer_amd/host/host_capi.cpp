// host_capi.cpp — extern "C" view of the C++ host layer, for ctypes (tests, bench.py).
// Scene functions need no GPU; renderer functions go through the C-ABI library.
#include <cstring>
#include <new>
#include <string>

#include "renderer.hpp"
#include "scene.hpp"

using namespace srt_host;

static thread_local std::string g_err;

extern "C" {

struct srt_host_scene {
    Scene scene;
    std::vector<srt_object> flat;
    std::string dump;
    explicit srt_host_scene(const char* path) : scene(path) {}
};

// Scene(file).Load()   (Raytracer/Scene.hpp:18-20,27-80)
srt_host_scene* srt_host_scene_load(const char* path) {
    srt_host_scene* s = new (std::nothrow) srt_host_scene(path ? path : "");
    if (!s) return nullptr;
    s->scene.Load();
    s->flat = s->scene.Flatten();
    return s;
}
srt_host_scene* srt_host_scene_new(const char* path) {
    return new (std::nothrow) srt_host_scene(path ? path : "");
}
void srt_host_scene_free(srt_host_scene* s) { delete s; }
size_t srt_host_scene_count(const srt_host_scene* s) { return s->scene.GetObjects().size(); }
// ObjectsToRender as the C-ABI array; valid until the scene is freed or modified
const srt_object* srt_host_scene_objects(srt_host_scene* s) {
    s->flat = s->scene.Flatten();
    return s->flat.data();
}
// EXTENSION: geometry of the scene's Mesh objects (index = srt_object.mesh); valid until the scene changes
size_t srt_host_scene_mesh_count(srt_host_scene* s) { return s->scene.MeshViews().size(); }
int srt_host_scene_mesh(srt_host_scene* s, size_t i, srt_mesh* out) {
    std::vector<srt_mesh> v = s->scene.MeshViews();
    if (i >= v.size()) return 1;
    *out = v[i];
    return 0;
}
const char* srt_host_scene_error(const srt_host_scene* s) { return s->scene.lastError().c_str(); }
const char* srt_host_scene_name(const srt_host_scene* s) { return s->scene.sceneName.c_str(); }
const char* srt_host_scene_object_name(const srt_host_scene* s, size_t i) {
    return i < s->scene.GetObjects().size() ? s->scene.GetObjects()[i].name.c_str() : "";
}
// Scene::AddObject (Scene.hpp:105-107)
void srt_host_scene_add(srt_host_scene* s, const srt_object* o, const char* name) {
    SceneObject so;
    so.type = (RendererType)o->type;
    so.name = name ? name : "";
    for (int i = 0; i < 3; ++i) {
        so.position[i] = o->position[i];
        so.size[i] = o->half_size[i];
    }
    so.radius = o->radius;
    so.material.Smoothness = o->material.smoothness;
    so.material.SpecularAmount = o->material.specular_amount;
    so.material.BaseColor = Color3(o->material.base_color[0], o->material.base_color[1], o->material.base_color[2]);
    so.material.EmissiveColor = Color3(o->material.emissive_color[0], o->material.emissive_color[1], o->material.emissive_color[2]);
    so.material.SpecularColor = Color3(o->material.specular_color[0], o->material.specular_color[1], o->material.specular_color[2]);
    s->scene.AddObject(so);
}
int srt_host_scene_remove(srt_host_scene* s, size_t index) { return s->scene.RemoveObject(index) ? 1 : 0; }
// Scene::SaveAs (Scene.hpp:101-104)
void srt_host_scene_save_as(srt_host_scene* s, const char* path) { s->scene.SaveAs(path); }
// the bytes Save() would write (dump(4)); valid until the next call
const char* srt_host_scene_dump(srt_host_scene* s) {
    s->dump = s->scene.Dump();
    return s->dump.c_str();
}
// Json::format_double, exposed for writer tests
// Json::parse + dump(indent) of an arbitrary document (differential tests against the reference's
// nlohmann/json).  Returns the length written (truncated to cap - 1), or (size_t)-1 on a parse error.
size_t srt_host_json_roundtrip(const char* text, int indent, char* out, size_t cap) {
    try {
        std::string t = Json::parse(text ? text : "").dump(indent);
        if (cap) {
            std::strncpy(out, t.c_str(), cap - 1);
            out[cap - 1] = 0;
        }
        return t.size();
    } catch (const std::exception& e) {
        g_err = e.what();
        return (size_t)-1;
    }
}

size_t srt_host_format_double(double v, char* out, size_t cap) {
    std::string t = Json::format_double(v);
    if (cap) {
        std::strncpy(out, t.c_str(), cap - 1);
        out[cap - 1] = 0;
    }
    return t.size();
}

// Transform::RotateAboutAxis (Common.hpp:287-291): basis = 9 floats right,up,forward
void srt_host_rotate_about_axis(float* right_up_forward, float angle, const float* axis) {
    Transform t;
    t.right = Vec3(right_up_forward[0], right_up_forward[1], right_up_forward[2]);
    t.up = Vec3(right_up_forward[3], right_up_forward[4], right_up_forward[5]);
    t.forward = Vec3(right_up_forward[6], right_up_forward[7], right_up_forward[8]);
    t.RotateAboutAxis(angle, Vec3(axis[0], axis[1], axis[2]));
    const Vec3* v[3] = {&t.right, &t.up, &t.forward};
    for (int i = 0; i < 3; ++i) {
        right_up_forward[3 * i] = v[i]->x;
        right_up_forward[3 * i + 1] = v[i]->y;
        right_up_forward[3 * i + 2] = v[i]->z;
    }
}

// ---- PathTraceRenderer -------------------------------------------------------------
struct srt_host_renderer {
    PathTraceRenderer* r = nullptr;
    std::string error;
};

const char* srt_host_last_error() { return g_err.c_str(); }

srt_host_renderer* srt_host_renderer_create(int device, int width, int height) {
    try {
        srt_host_renderer* h = new srt_host_renderer();
        h->r = new PathTraceRenderer(device, width, height);
        return h;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void srt_host_renderer_destroy(srt_host_renderer* h) {
    if (h) {
        delete h->r;
        delete h;
    }
}
#define SRT_HOST_TRY(h, body)                \
    try {                                    \
        body;                                \
        return 0;                            \
    } catch (const RendererError& e) {       \
        g_err = e.what();                    \
        return e.code();                     \
    } catch (const std::exception& e) {      \
        g_err = e.what();                    \
        return SRT_ERR_STATE;                \
    }
int srt_host_renderer_set_scene(srt_host_renderer* h, srt_host_scene* s) { SRT_HOST_TRY(h, h->r->SetScene(s->scene)) }
int srt_host_renderer_set_band(srt_host_renderer* h, int rb, int re) { SRT_HOST_TRY(h, h->r->SetRowBand(rb, re)) }
int srt_host_renderer_settings(srt_host_renderer* h, int fov, int max_bounces, int target_frames, uint32_t seed) {
    h->r->FOV = fov;
    h->r->MAXBOUNCES = max_bounces;
    h->r->TARGETFRAMES = target_frames;
    h->r->seed = seed;
    h->r->Invalidate();
    return 0;
}
// SIMPLEDRAW (:35), SCREEN_SCALE (:30), selectedObject (:53)
int srt_host_renderer_mode(srt_host_renderer* h, int simpledraw, float screen_scale, int selected_object) {
    h->r->SIMPLEDRAW = simpledraw != 0;
    h->r->SCREEN_SCALE = screen_scale;
    h->r->selectedObject = selected_object;
    h->r->Invalidate();
    return 0;
}
int srt_host_renderer_pick(srt_host_renderer* h, int mouse_x, int mouse_y, int* index) { SRT_HOST_TRY(h, *index = h->r->Pick(mouse_x, mouse_y)) }
int srt_host_renderer_set_camera(srt_host_renderer* h, const float* pos, const float* right_up_forward) {
    Transform& t = h->r->camera;
    t.position = Vec3(pos[0], pos[1], pos[2]);
    t.right = Vec3(right_up_forward[0], right_up_forward[1], right_up_forward[2]);
    t.up = Vec3(right_up_forward[3], right_up_forward[4], right_up_forward[5]);
    t.forward = Vec3(right_up_forward[6], right_up_forward[7], right_up_forward[8]);
    h->r->Invalidate();
    return 0;
}
void srt_host_renderer_invalidate(srt_host_renderer* h) { h->r->Invalidate(); }
// returns 1 if a frame was launched, 0 if ACCUMULATIONFRAMES == TARGETFRAMES, <0 on error
int srt_host_renderer_render_frame(srt_host_renderer* h) {
    try {
        return h->r->RenderFrame() ? 1 : 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
int srt_host_renderer_render_samples(srt_host_renderer* h, uint32_t count, int count_rays) {
    SRT_HOST_TRY(h, h->r->RenderSamples(count, count_rays != 0))
}
int srt_host_renderer_accumulation_frames(srt_host_renderer* h) { return h->r->ACCUMULATIONFRAMES; }
int srt_host_renderer_wait(srt_host_renderer* h) { SRT_HOST_TRY(h, h->r->Wait()) }
int srt_host_renderer_read_framebuffer(srt_host_renderer* h, void* dst, size_t pitch) { SRT_HOST_TRY(h, h->r->ReadFramebuffer(dst, pitch)) }
int srt_host_renderer_read_accumulator(srt_host_renderer* h, float* dst) {
    SRT_HOST_TRY(h, {
        std::vector<float> a = h->r->ReadAccumulator();
        std::memcpy(dst, a.data(), a.size() * sizeof(float));
    })
}
int srt_host_renderer_stats(srt_host_renderer* h, srt_stats* out) { SRT_HOST_TRY(h, *out = h->r->Stats()) }
void* srt_host_renderer_handle(srt_host_renderer* h) { return h->r->handle(); }

// ---- MultiGpuRenderer -----------------------------------------------------------------
struct srt_host_multi {
    MultiGpuRenderer* r = nullptr;
};
srt_host_multi* srt_host_multi_create(const int* devices, int n, int width, int height) {
    try {
        srt_host_multi* h = new srt_host_multi();
        h->r = new MultiGpuRenderer(std::vector<int>(devices, devices + n), width, height);
        return h;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void srt_host_multi_destroy(srt_host_multi* h) {
    if (h) {
        delete h->r;
        delete h;
    }
}
int srt_host_multi_set_scene(srt_host_multi* h, srt_host_scene* s) { SRT_HOST_TRY(h, h->r->SetScene(s->scene)) }
int srt_host_multi_configure(srt_host_multi* h, const float* pos, const float* right_up_forward, int fov, int max_bounces, uint32_t seed) {
    Transform t;
    t.position = Vec3(pos[0], pos[1], pos[2]);
    t.right = Vec3(right_up_forward[0], right_up_forward[1], right_up_forward[2]);
    t.up = Vec3(right_up_forward[3], right_up_forward[4], right_up_forward[5]);
    t.forward = Vec3(right_up_forward[6], right_up_forward[7], right_up_forward[8]);
    SRT_HOST_TRY(h, h->r->Configure(t, fov, max_bounces, seed))
}
int srt_host_multi_render_samples(srt_host_multi* h, uint32_t count, int count_rays) { SRT_HOST_TRY(h, h->r->RenderSamples(count, count_rays != 0)) }
int srt_host_multi_read_framebuffer(srt_host_multi* h, void* dst, size_t pitch) { SRT_HOST_TRY(h, h->r->ReadFramebuffer(dst, pitch)) }
int srt_host_multi_band(srt_host_multi* h, int i, int* begin, int* end) { SRT_HOST_TRY(h, h->r->Band((size_t)i, begin, end)) }
int srt_host_multi_balance(srt_host_multi* h) { SRT_HOST_TRY(h, h->r->BalanceBands()) }
int srt_host_multi_use_equal_bands(srt_host_multi* h, int equal) { SRT_HOST_TRY(h, h->r->UseEqualBands(equal != 0)) }
int srt_host_multi_use_manual_bands(srt_host_multi* h, int manual) { SRT_HOST_TRY(h, h->r->UseManualBands(manual != 0)) }
int srt_host_multi_set_auto_balance_min_samples(srt_host_multi* h, uint32_t per_device) { SRT_HOST_TRY(h, h->r->SetAutoBalanceMinSamples(per_device)) }
int srt_host_multi_set_row_band(srt_host_multi* h, int i, int begin, int end) { SRT_HOST_TRY(h, h->r->part((size_t)i).SetRowBand(begin, end)) }
int srt_host_multi_stats(srt_host_multi* h, srt_stats* out, int n) {
    SRT_HOST_TRY(h, {
        std::vector<srt_stats> st = h->r->Stats();
        for (int i = 0; i < n && i < (int)st.size(); ++i) out[i] = st[(size_t)i];
    })
}

}  // extern "C"
