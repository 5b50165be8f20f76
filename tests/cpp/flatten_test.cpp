// Runs the adapter's mesh walk (vermilion_amd/adapter/flatten.h, the template HipPathTracer.cpp instantiates with
// aiMesh*) on plain stand-in meshes and prints what it produced; tests/test_adapter_flatten.py checks the numbers.
// A stand-in pins no arithmetic (there is none in the walk): what runs here is the index order of
// MeshEngine::createBVH (meshEngine.cpp:659-718) and the rule for meshes without UVs.
#include <cstdio>
#include <vector>

#include "flatten.h"

struct V3 { float x, y, z; };
struct Face { unsigned mNumIndices; unsigned *mIndices; };
struct Mesh {
    unsigned mNumVertices = 0, mNumFaces = 0;
    V3 *mVertices = nullptr, *mNormals = nullptr;
    V3 *mTextureCoords[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    Face *mFaces = nullptr;
    bool HasTextureCoords(unsigned i) const { return mTextureCoords[i] != nullptr && mNumVertices > 0; }
};

// mesh m: nv vertices with position (100 m + i, 0.5, -i), normal (i, m, 1), uv (m + i / 16, i / 32) if with_uv;
// faces f = (f, (f + 2) % nv, (f + 1) % nv): deliberately not the identity order
static Mesh *make(int m, unsigned nv, unsigned nf, bool with_uv) {
    Mesh *me = new Mesh;
    me->mNumVertices = nv, me->mNumFaces = nf;
    me->mVertices = new V3[nv], me->mNormals = new V3[nv];
    if (with_uv) me->mTextureCoords[0] = new V3[nv];
    for (unsigned i = 0; i < nv; ++i) {
        me->mVertices[i] = {100.f * m + i, 0.5f, -(float)i};
        me->mNormals[i] = {(float)i, (float)m, 1.f};
        if (with_uv) me->mTextureCoords[0][i] = {m + i / 16.f, i / 32.f, 0.f};
    }
    me->mFaces = new Face[nf];
    for (unsigned f = 0; f < nf; ++f) {
        me->mFaces[f].mNumIndices = 3;
        me->mFaces[f].mIndices = new unsigned[3]{f % nv, (f + 2) % nv, (f + 1) % nv};
    }
    return me;
}

int main() {
    // UV mesh, no-UV mesh, UV mesh, no-UV mesh: both UvRule values differ on meshes 1 and 3 only
    std::vector<Mesh *> meshes = {make(0, 5, 3, true), make(1, 4, 2, false), make(2, 6, 4, true), make(3, 3, 1, false)};
    for (int rule = 0; rule < 2; ++rule) {
        std::vector<float> pos, nrm, uv;
        const size_t n = Vermilion::flattenMeshes(meshes, rule ? Vermilion::UvRule::CarryOverFromPreviousMesh : Vermilion::UvRule::Zero,
                                                  pos, nrm, uv);
        std::printf("rule %d tris %zu sizes %zu %zu %zu\n", rule, n, pos.size(), nrm.size(), uv.size());
        for (size_t t = 0; t < n; ++t) {
            std::printf("t %zu pos", t);
            for (int k = 0; k < 9; ++k) std::printf(" %.9g", pos[t * 9 + k]);
            std::printf(" nrm");
            for (int k = 0; k < 9; ++k) std::printf(" %.9g", nrm[t * 9 + k]);
            std::printf(" uv");
            for (int k = 0; k < 6; ++k) std::printf(" %.9g", uv[t * 6 + k]);
            std::printf("\n");
        }
    }
    // an empty scene and a mesh with no faces
    std::vector<Mesh *> none;
    std::vector<float> p, q, r;
    std::printf("empty %zu\n", Vermilion::flattenMeshes(none, Vermilion::UvRule::Zero, p, q, r));
    std::vector<Mesh *> hollow = {make(7, 3, 0, true)};
    std::printf("hollow %zu %zu\n", Vermilion::flattenMeshes(hollow, Vermilion::UvRule::Zero, p, q, r), p.size());
    return 0;
}
