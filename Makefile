# Top-level build: the HIP library (gfx950 only), the C host library / CLI, and the oracle.
#   make            -> fermi_amd/lib/libfmdhip.so  fermi_amd/lib/libfmdhost.so  oracle/liboracle.so
#   make ref        -> oracle/_ref/ (only where /root/reference exists)
HIPCC    ?= hipcc
ARCH     ?= gfx950
HIPFLAGS  = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value $(EXTRA)
CC        = gcc
CFLAGS    = -O2 -g -Wall -fPIC -std=gnu11

HIP_SRCS  = $(wildcard fermi_amd/csrc/*.hip)
HIP_HDRS  = $(wildcard fermi_amd/csrc/*.h) $(wildcard fermi_amd/csrc/*.inc) include/fmd_hip.h
HIP_OBJS  = $(patsubst fermi_amd/csrc/%.hip,build/%.o,$(HIP_SRCS))
HOST_SRCS = $(filter-out fermi_amd/host/main.c,$(wildcard fermi_amd/host/*.c))
HOST_HDRS = $(wildcard fermi_amd/host/*.h) include/fmd_hip.h

all: fermi_amd/lib/libfmdhip.so fermi_amd/lib/libfmdhip_count.so host cli oracle

build/%.o: fermi_amd/csrc/%.hip $(HIP_HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

fermi_amd/lib/libfmdhip.so: $(HIP_OBJS)
	@mkdir -p fermi_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(HIP_OBJS) -o $@

# the same sources with the gathers instrumented (fmd_count_lines, fmd_wave.h): bench.py runs one step of every
# leg through it to count the rank blocks the shipped kernels request (roofline.frac in DEVICE bytes)
CNT_OBJS  = $(patsubst fermi_amd/csrc/%.hip,build/count/%.o,$(HIP_SRCS))
build/count/%.o: fermi_amd/csrc/%.hip $(HIP_HDRS)
	@mkdir -p build/count
	$(HIPCC) $(HIPFLAGS) -DFMD_COUNT_LINES=1 -c $< -o $@
fermi_amd/lib/libfmdhip_count.so: $(CNT_OBJS)
	@mkdir -p fermi_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(CNT_OBJS) -o $@

host: fermi_amd/lib/libfmdhost.so
fermi_amd/lib/libfmdhost.so: $(HOST_SRCS) $(HOST_HDRS) fermi_amd/lib/libfmdhip.so
	@mkdir -p fermi_amd/lib
	$(CC) $(CFLAGS) -shared -Iinclude $(HOST_SRCS) -o $@ -Lfermi_amd/lib -lfmdhip -Wl,-rpath,'$$ORIGIN' -lpthread -lm -lz

cli: fermi_amd/bin/fermi-amd
fermi_amd/bin/fermi-amd: fermi_amd/host/main.c fermi_amd/lib/libfmdhost.so fermi_amd/lib/libfmdhip.so
	@mkdir -p fermi_amd/bin
	$(CC) $(CFLAGS) -Iinclude -Ifermi_amd/host fermi_amd/host/main.c -o $@ -Lfermi_amd/lib -lfmdhost -lfmdhip -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -s -C oracle oracle
ref:
	$(MAKE) -s -C oracle ref

# A/B variant of the HIP library: make variant NAME=nt EXTRA=-DFMD_GLDS_AUX=2  -> fermi_amd/lib/libfmdhip_nt.so
variant:
	@mkdir -p build/$(NAME)
	for f in $(HIP_SRCS); do $(HIPCC) $(HIPFLAGS) -c $$f -o build/$(NAME)/$$(basename $$f .hip).o || exit 1; done
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC build/$(NAME)/*.o -o fermi_amd/lib/libfmdhip_$(NAME).so

clean:
	rm -rf build fermi_amd/lib/*.so
	$(MAKE) -s -C oracle clean
.PHONY: all host cli oracle ref clean variant
