// scene.cpp — see scene.hpp. Reader/writer of the reference's scene JSON.
#include "scene.hpp"

#include <cmath>
#include <fstream>
#include <sstream>

namespace srt_host {

Json SceneObject::ToJSON() const {
    Json data;  // Object.hpp:27-43
    data["Name"] = name;
    data["Position"] = Json::array({position[0], position[1], position[2]});
    data["Material"]["Smoothness"] = material.Smoothness;
    data["Material"]["Metalness"] = material.SpecularAmount;  // written, never read back (:33)
    data["Material"]["Color"] = Json::array({material.BaseColor.r, material.BaseColor.g, material.BaseColor.b});
    data["Material"]["Emissive"] = Json::array({material.EmissiveColor.r, material.EmissiveColor.g, material.EmissiveColor.b});
    data["Material"]["SpecularColor"] = Json::array({material.SpecularColor.r, material.SpecularColor.g, material.SpecularColor.b});
    data["Material"]["SpecularAmount"] = material.SpecularAmount;
    Json renderer = Json::object();
    if (type == RendererType::Sphere) {  // Object.hpp:143-147
        renderer["Type"] = "Sphere";
        renderer["Radius"] = radius;
    } else if (type == RendererType::Cube) {  // :218-222
        renderer["Type"] = "Cube";
        renderer["Size"] = Json::array({size[0], size[1], size[2]});
    } else if (type == RendererType::Mesh) {  // EXTENSION
        renderer["Type"] = "Mesh";
        if (!meshPrimitive.empty()) {
            renderer["Primitive"] = meshPrimitive;
            renderer["Radius"] = radius;
            renderer["Stacks"] = meshStacks;
            renderer["Slices"] = meshSlices;
        } else {
            Json v = Json::array(), ix = Json::array();
            for (float f : meshVertices) v.push_back(Json(f));
            for (uint32_t k : meshIndices) ix.push_back(Json((int64_t)k));
            renderer["Vertices"] = v;
            renderer["Indices"] = ix;
        }
    } else {
        renderer["Type"] = "None";  // :41
    }
    data["Renderer"] = renderer;
    return data;
}

srt_object SceneObject::Flatten() const {
    srt_object o{};
    o.type = (int32_t)type;
    o.mesh = -1;
    for (int i = 0; i < 3; ++i) {
        o.position[i] = position[i];
        o.half_size[i] = size[i];
    }
    o.radius = radius;
    o.material.smoothness = material.Smoothness;
    o.material.specular_amount = material.SpecularAmount;
    const Color3 *b = &material.BaseColor, *e = &material.EmissiveColor, *s = &material.SpecularColor;
    o.material.base_color[0] = b->r, o.material.base_color[1] = b->g, o.material.base_color[2] = b->b;
    o.material.emissive_color[0] = e->r, o.material.emissive_color[1] = e->g, o.material.emissive_color[2] = e->b;
    o.material.specular_color[0] = s->r, o.material.specular_color[1] = s->g, o.material.specular_color[2] = s->b;
    return o;
}

static Color3 color_of(Json color) {  // Color(color[0], color[1], color[2])
    float r = color[0].as_float(), g = color[1].as_float(), b = color[2].as_float();
    return Color3(r, g, b);
}

void Scene::Load() {
    sceneObjects.clear();
    error_.clear();
    std::ifstream f(fileName, std::ios::binary);
    if (!f.good()) return;  // Scene.hpp:30-32
    try {
        std::stringstream ss;
        ss << f.rdbuf();
        Json data = Json::parse(ss.str());                 // :34
        sceneName = data["SceneName"].as_string();         // :35 (missing/non-string throws)
        Json sceneObject = data["SceneObjects"];           // :36
        if (sceneObject.is_null()) return;                 // begin() == end()
        if (!sceneObject.is_array()) throw JsonError("SceneObjects is not an array");
        for (const Json& element : sceneObject.items()) {  // :38
            Json value = element;
            Json position = value["Position"];             // :40
            SceneObject obj;
            Json rtype = value["Renderer"]["Type"];
            if (rtype.is_string() && rtype.as_string() == "Sphere") {  // :43-45
                obj.type = RendererType::Sphere;
                obj.radius = value["Renderer"]["Radius"].as_float();
                (void)position[0].as_float(), (void)position[1].as_float(), (void)position[2].as_float();
            } else if (rtype.is_string() && rtype.as_string() == "Cube") {  // :46-52
                obj.type = RendererType::Cube;
                obj.size[0] = value["Renderer"]["Size"][0].as_float();
                obj.size[1] = value["Renderer"]["Size"][1].as_float();
                obj.size[2] = value["Renderer"]["Size"][2].as_float();
            } else if (rtype.is_string() && rtype.as_string() == "Mesh") {  // EXTENSION
                obj.type = RendererType::Mesh;
                Json r = value["Renderer"];
                if (r.contains("Primitive")) {
                    obj.meshPrimitive = r["Primitive"].as_string();
                    if (obj.meshPrimitive != "UVSphere") throw JsonError("unknown mesh primitive " + obj.meshPrimitive);
                    obj.radius = r["Radius"].as_float();
                    obj.meshStacks = (int)r["Stacks"].as_double();
                    obj.meshSlices = (int)r["Slices"].as_double();
                    if (obj.meshStacks < 2 || obj.meshSlices < 3 || obj.meshStacks > 4096 || obj.meshSlices > 4096)
                        throw JsonError("UVSphere needs 2 <= Stacks <= 4096 and 3 <= Slices <= 4096");
                    MakeUVSphere(obj.radius, obj.meshStacks, obj.meshSlices, obj.meshVertices, obj.meshIndices);
                } else {
                    for (const Json& f : r["Vertices"].items()) obj.meshVertices.push_back(f.as_float());
                    for (const Json& k : r["Indices"].items()) {
                        double d = k.as_double();
                        if (!(d >= 0 && d < 4294967296.0)) throw JsonError("mesh index out of range");
                        obj.meshIndices.push_back((uint32_t)d);
                    }
                    if (obj.meshVertices.size() % 3 || obj.meshIndices.size() % 3) throw JsonError("mesh arrays must hold triples");
                }
            } else {
                obj.type = RendererType::None;  // :53-55 inert, still occupies a list slot
            }
            obj.position[0] = position[0].as_float();  // :57
            obj.position[1] = position[1].as_float();
            obj.position[2] = position[2].as_float();
            if (value.contains("Material")) {  // :59-69
                Json material = value["Material"];
                obj.material.Smoothness = material.contains("Smoothness") ? material["Smoothness"].as_float() : 0.5f;
                obj.material.SpecularAmount = material.contains("SpecularAmount") ? material["SpecularAmount"].as_float() : 0.1f;
                obj.material.SpecularColor = color_of(material.contains("SpecularColor") ? material["SpecularColor"] : Json::array({1, 1, 1}));
                obj.material.BaseColor = color_of(material.contains("Color") ? material["Color"] : Json::array({1, 1, 1}));
                obj.material.EmissiveColor = color_of(material.contains("Emissive") ? material["Emissive"] : Json::array({0, 0, 0}));
            }
            obj.name = value["Name"].as_string();  // :71 (missing -> throws, object dropped)
            sceneObjects.push_back(obj);           // :72
        }
    } catch (const std::exception& e) {
        error_ = e.what();  // :75-77 prints and keeps what was loaded
    }
}

std::string Scene::Dump() const {
    Json data;  // Scene.hpp:89-96
    data["SceneName"] = sceneName;
    data["SceneObjects"] = Json::array();
    for (size_t i = 0; i < sceneObjects.size(); ++i) data["SceneObjects"][i] = sceneObjects[i].ToJSON();
    return data.dump(4);
}

void Scene::Save() const {
    std::ofstream f(fileName, std::ios::binary);  // :97-99
    f << Dump();
}

bool Scene::RemoveObject(size_t index) {
    if (index >= sceneObjects.size()) return false;
    sceneObjects.erase(sceneObjects.begin() + (std::ptrdiff_t)index);
    return true;
}

std::vector<srt_object> Scene::Flatten() const {
    std::vector<srt_object> out;
    out.reserve(sceneObjects.size());
    int mesh = 0;
    for (const SceneObject& o : sceneObjects) {
        out.push_back(o.Flatten());
        if (o.type == RendererType::Mesh) out.back().mesh = mesh++;
    }
    return out;
}

std::vector<srt_mesh> Scene::MeshViews() const {
    std::vector<srt_mesh> out;
    for (const SceneObject& o : sceneObjects)
        if (o.type == RendererType::Mesh)
            out.push_back(srt_mesh{o.meshVertices.data(), o.meshVertices.size() / 3, o.meshIndices.data(), o.meshIndices.size() / 3});
    return out;
}

// north pole, (stacks-1) rings of `slices` vertices, south pole; pole bands are fans.
// Arithmetic in double, stored as float; counter-clockwise (outward) winding.
void MakeUVSphere(float radius, int stacks, int slices, std::vector<float>& vertices, std::vector<uint32_t>& indices) {
    vertices.clear();
    indices.clear();
    const double R = (double)radius, PI = 3.141592653589793;
    auto put = [&](double x, double y, double z) {
        vertices.push_back((float)x);
        vertices.push_back((float)y);
        vertices.push_back((float)z);
    };
    put(0.0, R, 0.0);
    for (int i = 1; i < stacks; ++i) {
        const double phi = PI * i / stacks, y = R * std::cos(phi), r = R * std::sin(phi);
        for (int j = 0; j < slices; ++j) {
            const double th = 2.0 * PI * j / slices;
            put(r * std::cos(th), y, r * std::sin(th));
        }
    }
    put(0.0, -R, 0.0);
    auto ring = [&](int i, int j) { return (uint32_t)(1 + (i - 1) * slices + (j % slices)); };
    const uint32_t south = (uint32_t)(vertices.size() / 3 - 1);
    auto tri = [&](uint32_t a, uint32_t b, uint32_t c) {
        indices.push_back(a);
        indices.push_back(b);
        indices.push_back(c);
    };
    for (int j = 0; j < slices; ++j) tri(0, ring(1, j + 1), ring(1, j));
    for (int i = 1; i < stacks - 1; ++i)
        for (int j = 0; j < slices; ++j) {
            const uint32_t a = ring(i, j), b = ring(i, j + 1), c = ring(i + 1, j), d = ring(i + 1, j + 1);
            tri(a, b, d);
            tri(a, d, c);
        }
    for (int j = 0; j < slices; ++j) tri(south, ring(stacks - 1, j), ring(stacks - 1, j + 1));
}

}  // namespace srt_host
