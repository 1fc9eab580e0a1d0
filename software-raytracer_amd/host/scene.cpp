// scene.cpp — see scene.hpp. Reader/writer of the reference's scene JSON.
#include "scene.hpp"

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
    } else {
        renderer["Type"] = "None";  // :41
    }
    data["Renderer"] = renderer;
    return data;
}

srt_object SceneObject::Flatten() const {
    srt_object o{};
    o.type = (int32_t)type;
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
    for (const SceneObject& o : sceneObjects) out.push_back(o.Flatten());
    return out;
}

}  // namespace srt_host
