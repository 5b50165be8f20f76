// Type-only stand-in for <assimp/scene.h> — see tests/stubs/README.md.  Member names follow
// Assimp's public structs as the reference uses them (core/engines/meshEngine.cpp:660-718).
#pragma once
struct aiVector3D {
    float x, y, z;
};
struct aiFace {
    unsigned int mNumIndices;
    unsigned int *mIndices;
};
struct aiMesh {
    unsigned int mNumVertices, mNumFaces;
    aiVector3D *mVertices;
    aiVector3D *mNormals;
    aiVector3D *mTextureCoords[8];
    aiFace *mFaces;
    unsigned int mMaterialIndex;
    bool HasTextureCoords(unsigned int index) const;
};
struct aiMaterialProperty {
    unsigned int mSemantic;
};
struct aiMaterial {
    unsigned int mNumProperties;
    aiMaterialProperty **mProperties;
};
struct aiTexture {};
struct aiLight {};
struct aiCamera {};
struct aiAnimation {};
struct aiScene {
    unsigned int mNumMeshes, mNumMaterials;
    aiMesh **mMeshes;
    aiMaterial **mMaterials;
};
