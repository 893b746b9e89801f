# Top-level build: host library (g++), HIP library (hipcc, gfx950), C++ benchmark driver, oracle.
#   make            -> everything
#   make host hip oracle benchmark
# Built artefacts stay in-tree (hisparse_amd/lib/, oracle/liboracle.so): they are git-ignored but
# travel to the GPU box with the gpurun snapshot.
ROOT      := $(abspath .)
INC       := -I$(ROOT)/include
LIBDIR    := $(ROOT)/hisparse_amd/lib
CSRC      := $(ROOT)/hisparse_amd/csrc
CXX       ?= g++
HIPCC     ?= /opt/rocm/bin/hipcc
CXXFLAGS  := -O3 -std=c++17 -fPIC -Wall -Wextra -pthread $(INC)
HIPFLAGS  := -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Wall $(INC)

HOST_HDRS := $(wildcard include/hisparse/*.h) include/hisparse_host.h
HIP_HDRS  := include/hisparse_hip.h $(wildcard $(CSRC)/*.h) include/hisparse/common.h

.PHONY: all host hip cpu oracle benchmark clean
all: host hip cpu oracle benchmark

host: $(LIBDIR)/libhisparse_host.so
hip: $(LIBDIR)/libhisparse_hip.so
cpu: $(LIBDIR)/libhisparse_cpu.so
oracle: oracle/liboracle.so
benchmark: $(LIBDIR)/benchmark

$(LIBDIR):
	mkdir -p $(LIBDIR)

$(LIBDIR)/libhisparse_host.so: $(CSRC)/host_capi.cpp $(HOST_HDRS) | $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $< -lz

HIP_SRCS  := $(CSRC)/hs_api.cpp $(CSRC)/tiles_capi.cpp $(CSRC)/stream_tiles.cpp $(CSRC)/bitmap_tiles.cpp $(CSRC)/spmv_kernels.hip $(CSRC)/spmv_bitmap.hip $(CSRC)/spmspv.hip $(CSRC)/spmm_bitmap.hip $(CSRC)/spmm_mfma.hip $(CSRC)/gpu_tiles.hip
$(LIBDIR)/libhisparse_hip.so: $(HIP_SRCS) $(HIP_HDRS) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRCS) -pthread

# the same C-ABI on host threads for machines without a GPU: a separate library a driver links INSTEAD (never a fallback of the HIP one)
$(LIBDIR)/libhisparse_cpu.so: $(CSRC)/cpu_backend.cpp $(HIP_HDRS) include/hisparse/q8_24.h | $(LIBDIR)
	$(CXX) $(CXXFLAGS) -ffp-contract=off -I$(CSRC) -shared -o $@ $<

# host-only translation unit, built with hipcc for the HIP runtime and RCCL headers (multi-GPU path: one context per device, ncclAllGather of y)
$(LIBDIR)/benchmark: $(CSRC)/benchmark.cpp $(HOST_HDRS) include/hisparse_hip.h $(LIBDIR)/libhisparse_hip.so $(LIBDIR)/libhisparse_host.so | $(LIBDIR)
	$(HIPCC) -O3 -std=c++17 -Wall -pthread $(INC) -o $@ $< -L$(LIBDIR) -lhisparse_hip -lhisparse_host -lz -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

oracle/liboracle.so: oracle/cpu_ref.c
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) oracle/liboracle.so oracle/_ref
