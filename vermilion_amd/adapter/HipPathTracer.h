// HipPathTracer.h — the reference-side adapter: a Vermilion::Integrator whose
// Render() runs on an MI355X through libvermilion_hip.so.
//
// Built as part of Vermilion (add to SOURCE_FILES in CMakeLists.txt and link
// vermilion_hip); it is NOT compiled in this repository's image because it
// includes the reference's own headers, which need GLM/Assimp/OpenImageIO.
//
// Replaces: Vermilion::PathTracer (core/integrators/integrators.h:23-26,
// core/integrators/pathtracer.cpp:200-328).  Installed exactly like it:
//     auto integrator = new Vermilion::HipPathTracer();   // main.cpp:61
//     rEng->assignIntegrator(integrator);                 // main.cpp:63
#pragma once
#include "integrators/integrators.h"  // Vermilion::Integrator, Camera, MeshEngine
#include "vermilion_hip.h"

namespace Vermilion {

class HipPathTracer : public Integrator {
   public:
    // seed: the reference seeds from std::random_device (pathtracer.cpp:231)
    explicit HipPathTracer(uint64_t seed = 1, int device = 0) : mSeed(seed), mDevice(device) {}
    ~HipPathTracer() override;
    void Render(std::vector<Vermilion::Camera *> &cameraList, MeshEngine *mEng) override;

   private:
    bool upload(MeshEngine *mEng);
    uint64_t mSeed;
    int mDevice;
    vmx_scene *mScene = nullptr;
    const MeshEngine *mUploadedFrom = nullptr;
    size_t mUploadedFaces = 0;
};

}  // namespace Vermilion
