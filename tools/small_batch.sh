#!/bin/bash
# small-batch efficiency A/B: early-stop frame and rank 0 of 8 of the sharded frame, for libraries under build/
for l in "$@"; do echo "== $l"; VMX_LIB=build/$l python tools/es_probe.py 2>/dev/null | grep wall | tail -1; VMX_LIB=build/$l STRIPE=4 python tools/shard_probe.py 2>/dev/null | grep "world 8" | cut -c1-80; done
