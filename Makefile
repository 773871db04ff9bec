# Build of the MI355X-native MinimalOptiX render path.
#   make            -> device layer (HIP, gfx950), host library + CLI, oracle, hostsim
#   make device     -> minimaloptix_amd/lib/libmoptix.so        (hipcc --offload-arch=gfx950)
#   make host       -> minimaloptix_amd/lib/libmoptix_host.so, minimaloptix_amd/lib/moptix_render
#   make oracle     -> oracle/liboracle.so                      (test infrastructure)
#   make hostsim    -> tests/hostsim/libhostsim.so              (test infrastructure)
#   make loopback   -> tests/rccl_loopback/librccl_loopback.so  (test infrastructure: N ranks on one GPU without RCCL)
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
LIBDIR   := minimaloptix_amd/lib
CSRC     := minimaloptix_amd/csrc
HOST     := minimaloptix_amd/host

# -ffp-contract=off: arithmetic contract AC4; explicit fmaf() where a fused op is specified
EXTRA    ?=
LIBNAME  ?= libmoptix.so
BUILD    ?= build
# -fno-slp-vectorize: the SLP vectoriser pairs binary32 operations into v_pk_* instructions, which need their operands in aligned
# register pairs; in kernels that live at the register limit the moves and the pairing constraints cost more than the packed
# issue saves (packet kernel on coffee: 100.4 ms against 105.5 ms per 64 spp).  Same arithmetic.
HIPFLAGS := $(EXTRA) --offload-arch=$(ARCH) -O3 -fno-slp-vectorize -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -include cstring
CXXFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno -mavx2 -mfma -Wall -Wno-unused-function -Wno-unknown-pragmas

DEV_SRCS := $(CSRC)/moptix_api.hip $(CSRC)/megakernel.hip $(CSRC)/queuekernel.hip $(CSRC)/queuekernel_lean.hip $(CSRC)/packetkernel.hip $(CSRC)/packetkernel_n128.hip $(CSRC)/drainkernel.hip $(CSRC)/lbvh.hip
DEV_OBJS := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(DEV_SRCS))
DEV_HDRS := $(wildcard $(CSRC)/*.h) include/moptix.h
HOST_SRCS := $(HOST)/obj_loader.cpp $(HOST)/scene_file.cpp $(HOST)/scenes.cpp $(HOST)/standin_scenes.cpp $(HOST)/image_read.cpp $(HOST)/jpeg_read.cpp \
             $(HOST)/image_io.cpp $(HOST)/minimal_optix.cpp $(HOST)/host_capi.cpp
HOST_OBJS := $(patsubst $(HOST)/%.cpp,build/host_%.o,$(HOST_SRCS))
HOST_HDRS := $(wildcard $(HOST)/*.h) $(wildcard $(CSRC)/pt_*.h) include/moptix.h include/moptix_host.h

all: device host oracle hostsim loopback

device: $(LIBDIR)/$(LIBNAME)
host: $(LIBDIR)/libmoptix_host.so $(LIBDIR)/moptix_render
oracle:
	$(MAKE) -C oracle -s
hostsim:
	$(MAKE) -C tests/hostsim -s
loopback:
	$(MAKE) -C tests/rccl_loopback -s

$(BUILD)/queuekernel_lean.o: $(CSRC)/queuekernel.hip
$(BUILD)/packetkernel_n128.o: $(CSRC)/packetkernel.hip
# LLVM's scheduling strategy, per file: "max-ilp" for the 64-byte-node packet kernels (coffee 312.5 -> 309.9 ms, glass knot -1.5 %) and the drain kernel
# (lone path 18.9 -> 18.3 us per bounce); the 128-byte-node packet kernels (dining room +2-4 %) and the queue kernels (random spheres +4 %) lose with it
# and keep the default.  Same arithmetic, same image bits.
$(BUILD)/packetkernel.o $(BUILD)/drainkernel.o: HIPFLAGS += -mllvm -amdgpu-sched-strategy=max-ilp
$(BUILD)/%.o: $(CSRC)/%.hip $(DEV_HDRS) Makefile
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/$(LIBNAME): $(DEV_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(DEV_OBJS) -ldl -Wl,-rpath,/opt/rocm/lib

build/host_%.o: $(HOST)/%.cpp $(HOST_HDRS) Makefile
	@mkdir -p build
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(LIBDIR)/libmoptix_host.so: $(HOST_OBJS) $(LIBDIR)/libmoptix.so
	$(CXX) -shared -fPIC -o $@ $(HOST_OBJS) -L$(LIBDIR) -lmoptix -Wl,-rpath,'$$ORIGIN'

$(LIBDIR)/moptix_render: $(HOST)/main.cpp $(LIBDIR)/libmoptix_host.so
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/main.cpp -L$(LIBDIR) -lmoptix_host -lmoptix -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

clean:
	rm -rf build $(LIBDIR)/*.so $(LIBDIR)/moptix_render oracle/liboracle.so tests/hostsim/libhostsim.so tests/rccl_loopback/librccl_loopback.so

.PHONY: all device host oracle hostsim loopback clean
