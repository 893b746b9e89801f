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

.PHONY: all host hip cpu oracle benchmark clean prof variant
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

# One object per translation unit (an edit of one kernel file recompiles that file only); objects live in build/ (git-ignored, and listed in
# .gpurunignore: the GPU box needs the libraries, not the objects).
HIP_UNITS := hs_api.cpp tiles_capi.cpp stream_tiles.cpp bitmap_tiles.cpp sweep_tiles.cpp spmv_kernels.hip spmv_bitmap.hip spmv_sweep.hip spmspv.hip spmm_bitmap.hip spmm_mfma.hip spmm_sweep.hip gpu_tiles.hip
OBJDIR    := $(ROOT)/build
HIP_OBJS  := $(addprefix $(OBJDIR)/prod/,$(addsuffix .o,$(HIP_UNITS)))
PROF_OBJS := $(addprefix $(OBJDIR)/prof/,$(addsuffix .o,$(HIP_UNITS)))
$(OBJDIR)/prod/%.o: $(CSRC)/% $(HIP_HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -x hip -c -o $@ $<
$(OBJDIR)/prof/%.o: $(CSRC)/% $(HIP_HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -DHISPARSE_PROFILING -x hip -c -o $@ $<
$(LIBDIR)/libhisparse_hip.so: $(HIP_OBJS) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_OBJS) -pthread

# The profiling library: the same sources with -DHISPARSE_PROFILING -- the HISPARSE_ABLATE / HISPARSE_DEPTH instantiations (most give WRONG
# results by design) and the timeline builds.  Never loaded by default; tools/ select it with HISPARSE_HIP_LIB.  `make HISPARSE_PROFILING=1`
# (or `make prof`) builds it next to the product library.
prof: $(LIBDIR)/libhisparse_hip_prof.so
$(LIBDIR)/libhisparse_hip_prof.so: $(PROF_OBJS) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(PROF_OBJS) -pthread
ifeq ($(HISPARSE_PROFILING),1)
all: prof
endif

# An A/B variant of the product library: make variant NAME=<name> DEFS=-D<switch>=<value> -> libhisparse_hip_<name>.so
# (tools select it with HISPARSE_HIP_LIB, like the profiling library)
ifdef NAME
VAR_OBJS := $(addprefix $(OBJDIR)/$(NAME)/,$(addsuffix .o,$(HIP_UNITS)))
$(OBJDIR)/$(NAME)/%.o: $(CSRC)/% $(HIP_HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) $(DEFS) -x hip -c -o $@ $<
variant: $(VAR_OBJS) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $(LIBDIR)/libhisparse_hip_$(NAME).so $(VAR_OBJS) -pthread
endif

# the same C-ABI on host threads for machines without a GPU: a separate library a driver links INSTEAD (never a fallback of the HIP one)
$(LIBDIR)/libhisparse_cpu.so: $(CSRC)/cpu_backend.cpp $(HIP_HDRS) include/hisparse/q8_24.h | $(LIBDIR)
	$(CXX) $(CXXFLAGS) -ffp-contract=off -I$(CSRC) -shared -o $@ $<

# host-only translation unit, built with hipcc for the HIP runtime and RCCL headers (multi-GPU path: one context per device, ncclAllGather of y)
$(LIBDIR)/benchmark: $(CSRC)/benchmark.cpp $(HOST_HDRS) include/hisparse_hip.h $(LIBDIR)/libhisparse_hip.so $(LIBDIR)/libhisparse_host.so | $(LIBDIR)
	$(HIPCC) -O3 -std=c++17 -Wall -pthread $(INC) -o $@ $< -L$(LIBDIR) -lhisparse_hip -lhisparse_host -lz -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

oracle/liboracle.so: oracle/cpu_ref.c
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) $(OBJDIR) oracle/liboracle.so oracle/_ref
