#!/bin/bash
# The load-time planners and host builders (stream_tiles.cpp, sweep_tiles.cpp, bitmap_tiles.cpp, tiles_capi.cpp) under AddressSanitizer +
# UndefinedBehaviorSanitizer (g++, no GPU): built into a scratch library with stand-ins for the device passes (gpu_stub.cpp: never called, the
# host builders run) and for the exports hisparse_amd.device binds, then tests/test_tiles_cpu.py + tests/test_sweep_cpu.py against it; and the
# host library (formatter, loaders, npz: host_capi.cpp over include/hisparse/*.h) the same way against its tests.
set -e
cd "$(dirname "$0")/../.."
out=${SANITIZE_DIR:-/tmp/hisparse_sanitize}; mkdir -p $out
python - "$out" <<'PY'
import re, sys
from hisparse_amd import device
tiles = set(re.findall(r'\b(hs_\w+)\s*\(', open('hisparse_amd/csrc/tiles_capi.cpp').read()))
with open(sys.argv[1] + '/exports_stub.c', 'w') as f:
    for n in device.EXPORTS:
        if n not in tiles:
            f.write(('const char* %s(void) { return "stub"; }\n' if n in ('hs_strerror', 'hs_last_error') else 'int %s(void) { return -1; }\n') % n)
PY
gcc -c -fPIC -o $out/exports_stub.o $out/exports_stub.c
g++ -O1 -g -std=c++17 -fPIC -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Ihisparse_amd/csrc -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
    -shared -o $out/libtiles_asan.so hisparse_amd/csrc/{tiles_capi,stream_tiles,bitmap_tiles,sweep_tiles}.cpp tools/sanitize/gpu_stub.cpp $out/exports_stub.o
PRE="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)"      # (libstdc++ too: the preloaded runtime must find __cxa_throw)
LD_PRELOAD="$PRE" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 HISPARSE_HIP_LIB=$out/libtiles_asan.so \
    python -m pytest tests/test_tiles_cpu.py tests/test_sweep_cpu.py -x -q
g++ -O1 -g -std=c++17 -fPIC -pthread -Iinclude -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $out/libhisparse_host_asan.so hisparse_amd/csrc/host_capi.cpp -lz
LD_PRELOAD="$PRE" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 HISPARSE_HOST_LIB=$out/libhisparse_host_asan.so \
    python -m pytest tests/test_host_format.py tests/test_npz_loader.py tests/test_formatter_goldens.py tests/test_golden_fixtures.py -x -q
