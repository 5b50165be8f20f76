// stand-in for <assimp/cimport.h> — see tests/stubs/README.md (nothing the adapter check needs)
#pragma once
