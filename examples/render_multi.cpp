// render_multi.cpp — the same C++ host as render_cornell.cpp (main.cpp:58-95 call order, C ABI only)
// with a DEVICE LIST: one process, one scene replica per listed device, interleaved 16-row stripes,
// device-to-device gather on the first device (vmx_multi_*, include/vermilion_hip.h).  A device may be
// listed more than once (rehearsal on a box with fewer GPUs); the picture does not depend on the list.
//
//   ./examples/render_multi out.ppm 256 256 64 [seed] [devices, e.g. 0,1,2,3,4,5,6,7]
//
// Writes a binary PPM the way Camera::saveFrame quantises (floor(x*255), core/camera/camera.cpp:150-170)
// and prints the statistics the reference cannot report (camera.h:101-102 are never incremented).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vermilion_hip.h"

namespace {

// the 8-triangle Cornell-like set of vermilion_amd/scenes.py: floor, back wall, block front + top
void quad(std::vector<float> &pos, std::vector<float> &nrm, std::vector<float> &uv, const float a[3], const float b[3],
          const float c[3], const float d[3], const float n[3]) {
    const float *tri[2][3] = {{a, b, c}, {a, c, d}};
    const float tuv[2][6] = {{0, 0, 1, 0, 1, 1}, {0, 0, 1, 1, 0, 1}};
    for (int t = 0; t < 2; ++t) {
        for (int v = 0; v < 3; ++v) {
            pos.insert(pos.end(), tri[t][v], tri[t][v] + 3);
            nrm.insert(nrm.end(), n, n + 3);
        }
        uv.insert(uv.end(), tuv[t], tuv[t] + 6);
    }
}

}  // namespace

int main(int argc, char **argv) {
    const char *out = argc > 1 ? argv[1] : "cornell.ppm";
    const uint32_t W = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 256, H = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 256;
    const uint32_t spp = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 64;
    const uint64_t seed = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 1;
    std::vector<int> devices;
    for (const char *p = argc > 6 ? argv[6] : "0"; *p;) {
        devices.push_back(std::atoi(p));
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
    }

    std::vector<float> pos, nrm, uv;
    const float up[3] = {0, 1, 0}, front[3] = {0, 0, 1};
    const float f0[3] = {-600, 1, 600}, f1[3] = {600, 1, 600}, f2[3] = {600, 1, -800}, f3[3] = {-600, 1, -800};
    quad(pos, nrm, uv, f0, f1, f2, f3, up);
    const float b0[3] = {-600, 1, -800}, b1[3] = {600, 1, -800}, b2[3] = {600, 900, -800}, b3[3] = {-600, 900, -800};
    quad(pos, nrm, uv, b0, b1, b2, b3, front);
    const float k0[3] = {-250, 1, 0}, k1[3] = {150, 1, 0}, k2[3] = {150, 400, 0}, k3[3] = {-250, 400, 0};
    quad(pos, nrm, uv, k0, k1, k2, k3, front);
    const float t0[3] = {-250, 400, 0}, t1[3] = {150, 400, 0}, t2[3] = {150, 400, -400}, t3[3] = {-250, 400, -400};
    quad(pos, nrm, uv, t0, t1, t2, t3, up);

    vmx_multi *scene = nullptr;
    if (vmx_multi_create(pos.data(), nrm.data(), uv.data(), (uint32_t)(pos.size() / 9), nullptr, 0, 4, VMX_BVH_REFERENCE,
                         devices.data(), (uint32_t)devices.size(), &scene) != VMX_OK) {
        std::fprintf(stderr, "scene: %s\n", vmx_last_error());
        return 1;
    }
    vmx_camera cam;
    std::memset(&cam, 0, sizeof(cam));
    cam.position[0] = 0, cam.position[1] = 420, cam.position[2] = 1900;
    cam.back_distance = 6.0f;  // renderEngine.cpp:135
    cam.back_size[0] = 3.6f, cam.back_size[1] = 3.6f * (float)H / (float)W;
    cam.image_res[0] = W, cam.image_res[1] = H;
    cam.rays_per_pixel = spp;
    vmx_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    opts.seed = seed;
    opts.early_stop = 1;
    opts.sampling = VMX_SAMPLING_PARITY;
    std::vector<float> frame((size_t)W * H * 5);
    vmx_stats st;
    if (vmx_multi_render(scene, &cam, &opts, frame.data(), &st) != VMX_OK) {
        std::fprintf(stderr, "render: %s\n", vmx_last_error());
        vmx_multi_destroy(scene);
        return 1;
    }
    FILE *f = std::fopen(out, "wb");
    if (!f) return 2;
    std::fprintf(f, "P6\n%u %u\n255\n", W, H);
    double sum = 0;
    for (size_t p = 0; p < (size_t)W * H; ++p) {
        unsigned char rgb[3];
        for (int c = 0; c < 3; ++c) {
            rgb[c] = (unsigned char)std::floor(frame[p * 5 + c] * 255.0f);
            sum += frame[p * 5 + c];
        }
        std::fwrite(rgb, 1, 3, f);
    }
    std::fclose(f);
    std::printf("%u device(s): ", vmx_multi_world(scene));
    std::printf("frame %ux%u spp %u seed %llu: rays %llu (+%llu secondary), samples %llu, %.2f ms device, checksum %.9g\n", W, H,
                spp, (unsigned long long)seed, (unsigned long long)st.rays_primary, (unsigned long long)st.rays_secondary,
                (unsigned long long)st.samples, st.ms_device, sum);
    vmx_multi_destroy(scene);
    return 0;
}
