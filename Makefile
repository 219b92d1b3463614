# Top-level build: everything is compiled for gfx950 (MI355X) only.
#
#   make            -> lib + cli + oracle
#   make lib        -> openmp-parallel-computing_amd/lib/libgrayhip.so   (HIP kernels + C ABI)
#   make cli        -> bin/grayscale, bin/grayscale_sobel                (drop-in CLIs, same argv/stdout/exit codes
#                                                                         as the reference's monolithic/bin/grayscale)
#   make oracle     -> oracle/liboracle.so (+ oracle/_ref/ when /root/reference is present) -- test infrastructure
#
# hipcc cross-compiles gfx950 without a GPU present.

PKG      := openmp-parallel-computing_amd
CSRC     := $(PKG)/csrc
LIBDIR   := $(PKG)/lib
LIB      := $(LIBDIR)/libgrayhip.so
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950

# -ffp-contract=off: the IEEE luma model needs un-fused mul/add (FMA mode calls fmaf explicitly).
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
            -Iinclude -I$(CSRC) -Wall -Wno-unused-function $(HIPFLAGS_EXTRA)
HIP_SRCS := $(CSRC)/gray.hip $(CSRC)/sobel.hip $(CSRC)/fused.hip $(CSRC)/conv3x3.hip $(CSRC)/reduce.hip $(CSRC)/old_api.hip $(CSRC)/streamed.hip $(CSRC)/comm.hip $(CSRC)/shim.hip
HIP_OBJS := $(HIP_SRCS:.hip=.o)
HIP_HDRS := include/grayhip.h $(CSRC)/gh_common.h $(CSRC)/sobel_tile.h $(CSRC)/stream_core.h

CXX      ?= g++
CXXFLAGS := -O2 -std=c++17 -Wall -pthread -Iinclude -I$(CSRC)

all: lib cli oracle

lib: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(HIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(HIP_OBJS) -ldl -o $@

CLI_COMMON := $(CSRC)/cli/image_io.cpp $(CSRC)/cli/jpeg_decode.cpp
CLI_CORE   := $(CSRC)/cli/cli_core.cpp $(CSRC)/cli/cli_core.h
cli: bin/grayscale bin/grayscale_sobel bin/grayhipd bin/gh_imgtool

bin/gh_imgtool: $(CSRC)/cli/imgtool.cpp $(CLI_COMMON) $(CSRC)/cli/image_io.h
	@mkdir -p bin
	$(CXX) $(CXXFLAGS) $(CSRC)/cli/imgtool.cpp $(CLI_COMMON) -lz -o $@

bin/grayscale: $(CSRC)/cli/main_gray.cpp $(CLI_COMMON) $(CLI_CORE) $(CSRC)/cli/image_io.h $(LIB)
	@mkdir -p bin
	$(CXX) $(CXXFLAGS) $(CSRC)/cli/main_gray.cpp $(CSRC)/cli/cli_core.cpp $(CLI_COMMON) -L$(LIBDIR) -lgrayhip -lz \
	    -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -o $@

bin/grayscale_sobel: $(CSRC)/cli/main_sobel.cpp $(CLI_COMMON) $(CLI_CORE) $(CSRC)/cli/image_io.h $(LIB)
	@mkdir -p bin
	$(CXX) $(CXXFLAGS) $(CSRC)/cli/main_sobel.cpp $(CSRC)/cli/cli_core.cpp $(CLI_COMMON) -L$(LIBDIR) -lgrayhip -lz \
	    -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -o $@

# resident server: the CLIs' work in one warm process (GRAYHIP_SERVER=<socket> turns the CLIs into its clients)
bin/grayhipd: $(CSRC)/cli/main_server.cpp $(CLI_COMMON) $(CLI_CORE) $(CSRC)/cli/image_io.h $(LIB)
	@mkdir -p bin
	$(CXX) $(CXXFLAGS) $(CSRC)/cli/main_server.cpp $(CSRC)/cli/cli_core.cpp $(CLI_COMMON) -L$(LIBDIR) -lgrayhip -lz \
	    -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -o $@

oracle:
	$(MAKE) -C oracle all

# developer build with the measurement probes (no-arithmetic skeletons, policy variants, handshake timestamps): never shipped,
# selected with GRAYHIP_LIB=build_probes/libgrayhip_probes.so
PROBE_OBJS := $(patsubst $(CSRC)/%.hip,build_probes/%.o,$(HIP_SRCS))
build_probes/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p build_probes
	$(HIPCC) $(HIPFLAGS) -DGH_ENABLE_PROBES -DGH_HALO_TRACE -c $< -o $@
probes: build_probes/libgrayhip_probes.so
build_probes/libgrayhip_probes.so: $(PROBE_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(PROBE_OBJS) -ldl -o $@

# device assembly of the LDS-DMA kernels with the library's own flags (tests/test_isa_vmcnt.py analyses it)
isa: build_isa/sobel.s build_isa/fused.s build_isa/conv3x3.s
build_isa/%.s: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p build_isa
	$(HIPCC) $(HIPFLAGS) --cuda-device-only -S $< -o $@ 2>/dev/null || $(HIPCC) $(HIPFLAGS) --cuda-device-only -S $< -o $@

clean:
	rm -rf build_isa
	rm -f $(HIP_OBJS) $(LIB) bin/grayscale bin/grayscale_sobel bin/grayhipd
	$(MAKE) -C oracle clean

.PHONY: all lib cli oracle clean isa probes

# developer micro-benchmark (kernel variants; not shipped)
ubench: bin/ubench
bin/ubench: tools/ubench.hip tools/ubench_sobel.h tools/ubench_valu.h tools/ubench_stream.h $(HIP_HDRS) $(LIB)
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Iinclude -I$(CSRC) tools/ubench.hip \
	    -L$(LIBDIR) -lgrayhip -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -o $@
.PHONY: ubench
