// stand-in for <assimp/postprocess.h> — see tests/stubs/README.md (nothing the adapter check needs)
#pragma once
