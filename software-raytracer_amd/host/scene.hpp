// scene.hpp — host-side mirror of the reference's scene model and JSON format.
//
// Mirrors (same names, argument meaning, error behaviour):
//   Scene             Raytracer/Scene.hpp:12-119   Load / Save / SaveAs / AddObject /
//                                                  RemoveObject / GetObjects / Unload
//   Object/Sphere/Box Raytracer/Object.hpp:19-43,86-102,143-147,170-206,218-222 (data + ToJSON)
//   Material          Raytracer/Common.hpp:293-319
// What is NOT mirrored: virtual Raytrace (the intersectors run on the GPU) and OnGUI.
//
// Wire format (Scene.hpp:27-104, Object.hpp:27-43):
//   {"SceneName": str, "SceneObjects": [ {"Name": str, "Position": [x,y,z],
//      "Material": {"Color":[3], "Emissive":[3], "Metalness": f, "Smoothness": f,
//                   "SpecularAmount": f, "SpecularColor":[3]},
//      "Renderer": {"Type":"Sphere","Radius":f} | {"Type":"Cube","Size":[3]} | {"Type":"None"}} ]}
#pragma once

#include <string>
#include <vector>

#include "json_min.hpp"
#include "srt_pathtrace.h"

namespace srt_host {

struct Color3 {  // r,g,b of the reference's Color; its 4-arg constructor clamps negatives (Common.hpp:253-257)
    float r = 0, g = 0, b = 0;
    Color3() = default;
    Color3(float r_, float g_, float b_) : r(r_ < 0 ? 0 : r_), g(g_ < 0 ? 0 : g_), b(b_ < 0 ? 0 : b_) {}
};

struct Material {  // Common.hpp:293-319
    float Smoothness = 0.5f;
    float SpecularAmount = 0.0f;
    Color3 BaseColor{1, 1, 1};
    Color3 EmissiveColor{0, 0, 0};
    Color3 SpecularColor{1, 1, 1};
};

enum class RendererType { None = SRT_OBJ_NONE, Sphere = SRT_OBJ_SPHERE, Cube = SRT_OBJ_BOX, Mesh = SRT_OBJ_MESH /* EXTENSION */ };

struct SceneObject {  // Object / Sphere / Box data members
    RendererType type = RendererType::None;
    std::string name;
    float position[3] = {0, 0, 0};  // transform.position
    float radius = 0;               // Sphere::radius
    float size[3] = {0, 0, 0};      // Box::size (half extents)
    Material material;
    // EXTENSION — "Renderer": {"Type": "Mesh", ...} (the reference's loader treats an unknown
    // type as an inert Object, Scene.hpp:53-55, so old readers degrade gracefully):
    //   explicit:    "Vertices": [x0,y0,z0, x1,...], "Indices": [i0,j0,k0, ...]
    //   procedural:  "Primitive": "UVSphere", "Radius": r, "Stacks": n, "Slices": m
    std::string meshPrimitive;      // "" = explicit arrays
    int meshStacks = 0, meshSlices = 0;
    std::vector<float> meshVertices;
    std::vector<uint32_t> meshIndices;

    Json ToJSON() const;            // Object.hpp:27-43 (+ :143-147 / :218-222)
    srt_object Flatten() const;     // -> C-ABI element (mesh index filled in by Scene::Flatten)
};

class Scene {
   public:
    std::string sceneName;
    explicit Scene(std::string file) : fileName(std::move(file)) {}

    const std::vector<SceneObject>& GetObjects() const { return sceneObjects; }
    std::vector<SceneObject>& Objects() { return sceneObjects; }
    const std::string& GetFilePath() const { return fileName; }

    // Scene.hpp:27-80.  Missing file -> silently empty scene (:30-32).  Any parse/type
    // error -> message kept in lastError(), objects loaded so far stay (:75-77).
    void Load();
    void Unload() { sceneObjects.clear(); }  // :81-87
    void Save() const;                       // :88-100, dump(4)
    void SaveAs(const std::string& file) {   // :101-104
        fileName = file;
        Save();
    }
    void AddObject(const SceneObject& o) { sceneObjects.push_back(o); }  // :105-107
    bool RemoveObject(size_t index);                                      // :108-115 (by identity there)

    std::string Dump() const;  // the exact bytes Save() writes
    // ObjectsToRender (Raytracer.cpp:61,293) as the C-ABI array, list order kept; mesh objects
    // get mesh = their ordinal among mesh objects, matching MeshViews()
    std::vector<srt_object> Flatten() const;
    std::vector<srt_mesh> MeshViews() const;  // pointers into this scene's objects
    const std::string& lastError() const { return error_; }

   private:
    std::string fileName;
    std::vector<SceneObject> sceneObjects;
    std::string error_;
};

// Deterministic latitude/longitude tessellation (BASELINE config 4: 224 x 224 -> 99,904 triangles)
void MakeUVSphere(float radius, int stacks, int slices, std::vector<float>& vertices, std::vector<uint32_t>& indices);

}  // namespace srt_host
