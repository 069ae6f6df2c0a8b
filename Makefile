# Builds the gfx950 shared library and the host-side checkers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
LIB   := mfcc_amd/lib/libmfcc_amd.so
SRC   := mfcc_amd/csrc/mfcc_amd.hip
HDRS  := $(wildcard mfcc_amd/csrc/*.h) include/mfcc_amd.h
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -fno-signed-zeros -fno-slp-vectorize -Wall -Wno-unused-function

all: $(LIB) host_sim

$(LIB): $(SRC) $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -o $@ $(SRC)

host_sim: tests/host_sim/fft_sim

tests/host_sim/fft_sim: tests/host_sim/fft_sim.cpp mfcc_amd/csrc/fft_lane.h mfcc_amd/csrc/tables.h
	g++ -O2 -std=c++17 -I mfcc_amd/csrc -o $@ $<

# experimental builds for A/B timing: make variant NAME=noslp EXTRA="-fno-slp-vectorize"
variant: $(SRC) $(HDRS)
	@mkdir -p build/variants
	$(HIPCC) $(HIPFLAGS) $(EXTRA) -o build/variants/libmfcc_$(NAME).so $(SRC)

# the experimental eight-wave kernel (mfcc_amd/csrc/split_kernel.h) as a library of its own, for
# tools/split_check.py and tests/test_gpu_split_variant.py
split: build/variants/libmfcc_split.so
build/variants/libmfcc_split.so: $(SRC) $(HDRS)
	@mkdir -p build/variants
	$(HIPCC) $(HIPFLAGS) -DMFCC_USE_SPLIT -o $@ $(SRC)

asm: $(SRC) $(HDRS)
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fno-signed-zeros -fno-slp-vectorize $(EXTRA) -S --cuda-device-only -o build/mfcc_amd.s $(SRC) -Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt

clean:
	rm -rf mfcc_amd/lib build tests/host_sim/fft_sim

.PHONY: all host_sim asm clean variant split
