// Type-only stand-in for <assimp/Importer.hpp> — see tests/stubs/README.md
#pragma once
#include "scene.h"
namespace Assimp {
class Importer {
   public:
    Importer();
    ~Importer();
};
}  // namespace Assimp
