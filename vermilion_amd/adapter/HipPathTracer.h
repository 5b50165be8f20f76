// HipPathTracer.h — the reference-side adapters: Vermilion::Integrator subclasses whose
// Render() runs on an MI355X through libvermilion_hip.so.
//
// Built as part of Vermilion (add to SOURCE_FILES in CMakeLists.txt and link
// vermilion_hip).  This repository's image has no GLM/Assimp/OpenImageIO, so here the file only goes
// through a syntax + type check against the reference's own headers (tests/test_adapter_compiles.py).
//
// HipPathTracer replaces Vermilion::PathTracer (core/integrators/integrators.h:23-26,
// core/integrators/pathtracer.cpp:200-328).  Installed exactly like it:
//     auto integrator = new Vermilion::HipPathTracer();   // main.cpp:61
//     rEng->assignIntegrator(integrator);                 // main.cpp:63
// or, for every GPU of the node from the one process main.cpp is:
//     auto integrator = new Vermilion::HipPathTracer(/*seed=*/1, std::vector<int>{0, 1, 2, 3, 4, 5, 6, 7});
// (the device-list form — vmx_multi_*, peer copies between DIFFERENT devices — has been rehearsed with every entry
// on one GPU only: no box with two GPUs was available to this repository's test runs.  tests/test_gpu_multi.py::
// test_two_physical_devices_equal_one and `bench.py --multi 0,1,...` exercise it the first time one is.)
#pragma once
#include <utility>
#include <vector>

#include "flatten.h"                  // flattenMeshes, UvRule / kUvOfMeshesWithoutUvs
#include "integrators/integrators.h"  // Vermilion::Integrator, Camera, MeshEngine
#include "vermilion_hip.h"

namespace Vermilion {

// what both adapters share: the device copy of MeshEngine's triangles / textures, the camera marshalling
// and the pixel write-back through Camera::setPixelValue
class HipIntegratorBase : public Integrator {
   public:
    // seed: the reference seeds from std::random_device (pathtracer.cpp:231) / time(0) (integrators.cpp:30)
    explicit HipIntegratorBase(uint64_t seed, std::vector<int> devices) : mSeed(seed), mDevices(std::move(devices)) {
        if (mDevices.empty()) mDevices.push_back(0);
    }
    ~HipIntegratorBase() override;
    // tree the library builds for the scene: VMX_BVH_REFERENCE (default: BVH::build's topology, the only one with
    // the reference's triangle-ID / tie order), VMX_BVH_SAH, VMX_BVH_LBVH or VMX_BVH_PLOC (a quality tree built on
    // the GPU in a few ms: 2.4x fewer node visits per ray).  Takes effect at the next upload.
    void setBuilder(uint32_t builder) {
        mBuilder = builder;
        mUploadedFrom = nullptr;
    }

   protected:
    bool upload(MeshEngine *mEng);
    static vmx_camera describe(const Camera *cam);
    static void writeBack(Camera *cam, const std::vector<float> &frame);
    // one frame through the library: the whole node when several devices were listed (vmx_multi_*:
    // one process, stripes sharded over the devices, gathered over xGMI on the first), else one device
    int renderFrame(const vmx_camera &c, const vmx_opts &o, bool bruteForce, uint32_t flags, float *frame, vmx_stats *st);
    uint64_t mSeed;
    std::vector<int> mDevices;
    uint32_t mBuilder = VMX_BVH_REFERENCE;
    vmx_scene *mScene = nullptr;   // one device
    vmx_multi *mMulti = nullptr;   // several
    const MeshEngine *mUploadedFrom = nullptr;
    size_t mUploadedFaces = 0, mUploadedTextures = 0;
};

class HipPathTracer : public HipIntegratorBase {
   public:
    explicit HipPathTracer(uint64_t seed = 1, int device = 0) : HipIntegratorBase(seed, {device}) {}
    HipPathTracer(uint64_t seed, std::vector<int> devices) : HipIntegratorBase(seed, std::move(devices)) {}
    void Render(std::vector<Vermilion::Camera *> &cameraList, MeshEngine *mEng) override;
    // vmx_opts.sampling of the frames.  Default: VMX_SAMPLING_PARITY — every ray the reference traces is traced, and
    // Camera::uRaysFired counts them.  setSampling(VMX_SAMPLING_PARITY | VMX_SAMPLING_ELIDE_DEAD) is the opt-in the
    // header describes: all Render leaves behind is Camera::mImage (r, g, b, sample count), which that flag keeps bit
    // for bit while the frame takes half the time; uRaysFired then counts only the rays that were traced (~22 %).
    void setSampling(uint32_t sampling) { mSampling = sampling; }

   private:
    uint32_t mSampling = VMX_SAMPLING_PARITY;
};

// Replaces Vermilion::BruteForceTracer (core/integrators/integrators.h:18-21,
// core/integrators/integrators.cpp:9-186), the integrator RenderEngine::Initialise installs when
// none is assigned (core/engines/renderEngine.cpp:49-53).
class HipBruteForceTracer : public HipIntegratorBase {
   public:
    // absAsInt: read the unqualified abs() of integrators.cpp:170 as C's abs(int) (VMX_BF_ABS_INT)
    explicit HipBruteForceTracer(uint64_t seed = 1, int device = 0, bool absAsInt = false)
        : HipIntegratorBase(seed, {device}), mFlags(absAsInt ? VMX_BF_ABS_INT : 0u) {}
    HipBruteForceTracer(uint64_t seed, std::vector<int> devices, bool absAsInt = false)
        : HipIntegratorBase(seed, std::move(devices)), mFlags(absAsInt ? VMX_BF_ABS_INT : 0u) {}
    void Render(std::vector<Vermilion::Camera *> &cameraList, MeshEngine *mEng) override;

   private:
    uint32_t mFlags;
};

}  // namespace Vermilion
