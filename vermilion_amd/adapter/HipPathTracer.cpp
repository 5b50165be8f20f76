// HipPathTracer.cpp — see HipPathTracer.h.  Marshal the reference's objects
// into the plain-C ABI of include/vermilion_hip.h; no rendering happens here.
#include "HipPathTracer.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace Vermilion {

HipIntegratorBase::~HipIntegratorBase() {
    vmx_scene_destroy(mScene);
    vmx_multi_destroy(mMulti);
}

// Flatten MeshEngine::sceneMeshes exactly as MeshEngine::createBVH does
// (core/engines/meshEngine.cpp:660-718): mesh-major, face-minor; that push
// order defines the triangle IDs.  BVH internals are private (bvh.h:19-26),
// so the library builds its own BVH with the same topology.
bool HipIntegratorBase::upload(MeshEngine *mEng) {
    std::vector<float> pos, nrm, uv;
    size_t faces = 0;
    for (aiMesh *mesh : mEng->sceneMeshes) faces += mesh->mNumFaces;
    if ((mScene || mMulti) && mUploadedFrom == mEng && mUploadedFaces == faces &&
        mUploadedTextures == mEng->boundTextures.size())
        return true;
    // the walk itself is a template over the mesh type (flatten.h) so that it also runs in this repository's CPU
    // tests with a plain stand-in mesh: mesh-major, face-minor, the face's three indices, kUvOfMeshesWithoutUvs
    flattenMeshes(mEng->sceneMeshes, kUvOfMeshesWithoutUvs, pos, nrm, uv);
    vmx_scene_destroy(mScene);
    vmx_multi_destroy(mMulti);
    mScene = nullptr, mMulti = nullptr;
    // NULL,0 -> the reference's eight hard-coded spheres (meshEngine.cpp:377-500); leaf size 4 (bvh.h:29)
    const int rc = mDevices.size() > 1
                       ? vmx_multi_create(pos.data(), nrm.data(), uv.data(), (uint32_t)faces, nullptr, 0, 4, mBuilder,
                                          mDevices.data(), (uint32_t)mDevices.size(), &mMulti)
                       : vmx_scene_create_ex(pos.data(), nrm.data(), uv.data(), (uint32_t)faces, nullptr, 0, 4, mBuilder,
                                             mDevices[0], &mScene);
    if (rc != VMX_OK) {
        std::fprintf(stderr, "vermilion_hip: %s\n", vmx_last_error());
        return false;
    }
    // boundTextures[0] is the only texture Radiance samples (pathtracer.cpp:63-66); BruteForceTracer also
    // reads boundTextures[1] (integrators.cpp:141-147).  Bound in order, as MeshEngine::bindTexture appends.
    for (size_t t = 0; t < mEng->boundTextures.size() && t < 2; ++t) {
        const VermiTexture &tx = mEng->boundTextures[t];
        if ((mMulti ? vmx_multi_bind_texture(mMulti, tx.pData, tx.nWidth, tx.nHeight, tx.nChannels)
                    : vmx_scene_bind_texture(mScene, tx.pData, tx.nWidth, tx.nHeight, tx.nChannels)) != VMX_OK)
            std::fprintf(stderr, "vermilion_hip: %s\n", vmx_last_error());
    }
    mUploadedFrom = mEng;
    mUploadedFaces = faces;
    mUploadedTextures = mEng->boundTextures.size();
    return true;
}

int HipIntegratorBase::renderFrame(const vmx_camera &c, const vmx_opts &o, bool bruteForce, uint32_t flags, float *frame,
                                   vmx_stats *st) {
    if (mMulti)
        return bruteForce ? vmx_multi_render_bruteforce(mMulti, &c, &o, flags, frame, st) : vmx_multi_render(mMulti, &c, &o, frame, st);
    return bruteForce ? vmx_render_bruteforce(mScene, &c, &o, flags, frame, st) : vmx_render(mScene, &c, &o, frame, st);
}

vmx_camera HipIntegratorBase::describe(const Camera *cam) {
    vmx_camera c;
    std::memset(&c, 0, sizeof(c));
    c.position[0] = cam->mPosition.x, c.position[1] = cam->mPosition.y, c.position[2] = cam->mPosition.z;
    // Camera::mRotation is what PathTracer::Render hands to glm::rotate (pathtracer.cpp:219-221): radians, x and y
    // already negated by the ctor (camera.cpp:43-47) — or whatever a host wrote into the public field.  Passed through
    // as it stands (a detour through degrees moved 9 % of arbitrary radians by one ulp)
    c.rotation_units = VMX_ROTATION_RADIANS;
    c.rotation_rad[0] = cam->mRotation.x, c.rotation_rad[1] = cam->mRotation.y, c.rotation_rad[2] = cam->mRotation.z;
    c.back_distance = cam->mDistToFilm;
    c.back_size[0] = cam->sensorSizeX, c.back_size[1] = cam->sensorSizeY;
    c.image_res[0] = cam->uImageU, c.image_res[1] = cam->uImageV;
    c.rays_per_pixel = cam->uSamplesPerPixel;
    return c;
}

void HipIntegratorBase::writeBack(Camera *cam, const std::vector<float> &frame) {
    pixelValue pv;                                           // camera.h:49-58
    pv.light = 0.f;
    for (uint64_t p = 0; p < cam->RenderTargetSize; ++p) {   // works for every renderMode (camera.cpp:88-124)
        const float *s = &frame[p * 5];
        pv.pixel = p, pv.red = s[0], pv.green = s[1], pv.blue = s[2], pv.alpha = s[3], pv.depth = s[4];
        cam->setPixelValue(pv);
    }
}

void HipPathTracer::Render(std::vector<Vermilion::Camera *> &cameraList, MeshEngine *mEng) {
    if (!mEng || !upload(mEng)) return;  // Render has no error channel (integrators.h:15): log and leave mImage untouched
    for (Camera *cam : cameraList) {     // pathtracer.cpp:210
        const vmx_camera c = describe(cam);
        vmx_opts o;
        std::memset(&o, 0, sizeof(o));
        o.seed = mSeed;
        o.early_stop = 1;                    // pathtracer.cpp:290-311
        o.sampling = mSampling;              // default: r2 = 10*U (pathtracer.cpp:156), rays without effect not traced
        std::vector<float> frame((size_t)cam->RenderTargetSize * 5);
        vmx_stats st;
        if (renderFrame(c, o, false, 0, frame.data(), &st) != VMX_OK) {
            std::fprintf(stderr, "HipPathTracer: %s\n", vmx_last_error());
            continue;
        }
        cam->uRaysFired = st.rays_primary + st.rays_secondary;  // camera.h:101 (the reference never fills it)
        writeBack(cam, frame);
    }
}

void HipBruteForceTracer::Render(std::vector<Vermilion::Camera *> &cameraList, MeshEngine *mEng) {
    if (!mEng || !upload(mEng)) return;
    for (Camera *cam : cameraList) {  // integrators.cpp:21
        const vmx_camera c = describe(cam);
        vmx_opts o;
        std::memset(&o, 0, sizeof(o));
        o.seed = mSeed;
        std::vector<float> frame((size_t)cam->RenderTargetSize * 5);
        vmx_stats st;
        if (renderFrame(c, o, true, mFlags, frame.data(), &st) != VMX_OK) {
            std::fprintf(stderr, "HipBruteForceTracer: %s\n", vmx_last_error());
            continue;
        }
        cam->uRaysFired = st.rays_primary + st.rays_secondary;
        writeBack(cam, frame);  // red/green/blue clamped, alpha = hit fraction, depth = last hitDistance (integrators.cpp:176-183)
    }
}

}  // namespace Vermilion
