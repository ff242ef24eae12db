#!/bin/bash
# Development: link a copy of the library whose tower kernel is compiled with extra -D switches (timing experiments).
#   scripts/build_variant.sh NAME -DTW_DEV_NO_VECTOR ...   ->  crazyara_amd/lib/variants/NAME.so
# every device compile takes the library's flags (no packed f32 arithmetic: crazyara_amd/build.py)
FLAGS=$(cd "$(dirname "$0")/.." && python3 -c 'from crazyara_amd import build; print(*build.device_flags())')
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -c "from crazyara_amd import build; build.build()" >/dev/null
mkdir -p crazyara_amd/lib/variants
obj=crazyara_amd/build/variant_$name.o
(cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $FLAGS -DCRA_DEVELOPMENT "$@" -x hip -c /root/repo/crazyara_amd/csrc/nn/tower.hip -o /root/repo/$obj)
objs=$(ls crazyara_amd/build/*.o | grep -v variant_ | grep -v nn_tower.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o crazyara_amd/lib/variants/$name.so $objs $obj -lpthread
echo built crazyara_amd/lib/variants/$name.so
