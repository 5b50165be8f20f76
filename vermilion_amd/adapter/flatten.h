// flatten.h — MeshEngine::sceneMeshes -> the plain float arrays of vmx_scene_create, in createBVH's push order.
//
// MeshEngine::createBVH (core/engines/meshEngine.cpp:660-718) walks sceneMeshes mesh by mesh and, inside a mesh, face
// by face, pushing one Triangle per face built from mVertices / mNormals / mTextureCoords[0] at the face's three
// indices; that push order defines the triangle IDs (`BVH::build` permutes a copy, bvh.cpp:252).  The same walk here
// fills pos / nrm (9 floats per triangle: v0, v1, v2) and uv (6 floats: uv0, uv1, uv2).
//
// A template over the mesh type so that the walk — the index order and the rule for meshes without UVs — also RUNS in
// this repository's CPU tests with a plain stand-in mesh (tests/cpp/flatten_test.cpp; stand-in types pin no arithmetic,
// there is none here): inside Vermilion it is instantiated with aiMesh* (HipPathTracer.cpp: upload).  A Mesh needs
//   mNumFaces, mFaces[f].mIndices[c], mVertices[i].{x,y,z}, mNormals[i].{x,y,z}, HasTextureCoords(0),
//   mTextureCoords[0][i].{x,y}
// which is what createBVH itself reads (meshEngine.cpp:669-714).
#pragma once
#include <cstddef>
#include <cstring>
#include <vector>

namespace Vermilion {

// UVs of a mesh WITHOUT texture coordinates.  createBVH (meshEngine.cpp:663-667) declares
// `glm::vec2 v0uv, v1uv, v2uv;` inside its per-mesh loop and assigns them only under HasTextureCoords(0), so for such a
// mesh the reference passes default-constructed vec2s to Triangle — and what those hold is the GLM version's choice:
// zeros with GLM <= 0.9.8 (or GLM_FORCE_CTOR_INIT), indeterminate with GLM 0.9.9's default (in practice the stack
// slots still hold the previous mesh's last face).  extern/glm is an unpinned submodule (.gitmodules:7-9): neither
// reading can be pinned.  It matters only for mixed UV / no-UV scenes with a bound texture.  One switch:
enum class UvRule { Zero, CarryOverFromPreviousMesh };
constexpr UvRule kUvOfMeshesWithoutUvs = UvRule::Zero;

// Appends every face of every mesh of `meshes` (any range of pointers to Mesh) to pos / nrm / uv; returns the number
// of triangles appended.
template <class MeshRange>
size_t flattenMeshes(const MeshRange &meshes, UvRule rule, std::vector<float> &pos, std::vector<float> &nrm,
                     std::vector<float> &uv) {
    size_t faces = 0;
    for (const auto *mesh : meshes) faces += mesh->mNumFaces;
    pos.reserve(pos.size() + faces * 9), nrm.reserve(nrm.size() + faces * 9), uv.reserve(uv.size() + faces * 6);
    float lastUv[6] = {0, 0, 0, 0, 0, 0};
    for (const auto *mesh : meshes) {  // meshEngine.cpp:660: mesh-major
        // a mesh WITHOUT UVs gets zeros, or whatever the previous mesh's last face left behind (UvRule)
        if (rule == UvRule::Zero) std::memset(lastUv, 0, sizeof(lastUv));
        for (unsigned f = 0; f < mesh->mNumFaces; ++f) {  // :669: face-minor
            for (int c = 0; c < 3; ++c) {                 // :673-675 / :685-687 / :696-698: the face's three indices
                const unsigned i = mesh->mFaces[f].mIndices[c];
                pos.push_back(mesh->mVertices[i].x), pos.push_back(mesh->mVertices[i].y), pos.push_back(mesh->mVertices[i].z);
                nrm.push_back(mesh->mNormals[i].x), nrm.push_back(mesh->mNormals[i].y), nrm.push_back(mesh->mNormals[i].z);
                if (mesh->HasTextureCoords(0)) {  // :700-709
                    lastUv[c * 2] = mesh->mTextureCoords[0][i].x;
                    lastUv[c * 2 + 1] = mesh->mTextureCoords[0][i].y;
                }
            }
            uv.insert(uv.end(), lastUv, lastUv + 6);
        }
    }
    return faces;
}

}  // namespace Vermilion
