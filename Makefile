# Build the MI355X-native library (gfx950 only) and the CPU oracle.
#   make -j8        -> rcognita_amd/lib/librcg.so  +  oracle/_build/liboracle.so
#   make lib        -> HIP library only (hipcc cross-compiles without a GPU)
#   make oracle     -> C oracle only (gcc)
#   make dev        -> rcognita_amd/lib/librcg_dev.so: the same sources with -DRCG_DEV (timing-only switches RCG_DBG of
#                      k_actor_dma compiled in; for tools/ only - a tool binds it with rcognita_amd._native.use_library(path))
#   make asan       -> build/asan/abi_asan: the C oracle and the HOST side of librcg (host-only compile of the .hip units,
#                      no device code) under clang -fsanitize=address,undefined, with tests/asan_driver.c; CPU only
MAKEFLAGS += -r
.SUFFIXES:
HIPCC   ?= hipcc
CC      := gcc
ARCH    ?= gfx950
ROOT    := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
CSRC    := $(ROOT)rcognita_amd/csrc
LIBDIR  := $(ROOT)rcognita_amd/lib
OBJDIR  := $(ROOT)build/obj
ORACLE  := $(ROOT)oracle

HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function -ffp-contract=fast \
            -I$(ROOT)include -MMD -MP
# the C ABI; the system-templated launchers and kernels: rcg_sys_inst.hip compiled once per (system, part), see that file
UNITS   := rcg_api
SYSP    := $(foreach s,Sys3WRobot.kVt3WRobot Sys3WRobotNI.kVt3WRobotNI Sys2Tank.kVt2Tank,$(foreach p,0 1 2 3 4,$(s).$(p)))
SYSFLAGS = -DRCG_SYS=$(word 1,$(subst ., ,$*)) -DRCG_SYS_VT=$(word 2,$(subst ., ,$*)) -DRCG_SYS_PART=$(word 3,$(subst ., ,$*))
# k_actor_dma instances: rcg_dma_inst.hip compiled once per (system, element type, group), see that file
DMA     := $(foreach s,Sys3WRobot Sys3WRobotNI Sys2Tank,$(foreach r,float double,$(foreach g,0 1 2 3 4 5 6 7,$(s).$(r).$(g))))
DMAFLAGS = -DRCG_INST_SYS=$(word 1,$(subst ., ,$*)) -DRCG_INST_REAL=$(word 2,$(subst ., ,$*)) \
           -DRCG_INST_GROUP=$(word 3,$(subst ., ,$*))
objs     = $(addprefix $(1)/rcg_sys.,$(addsuffix .o,$(SYSP))) $(addprefix $(1)/,$(addsuffix .o,$(UNITS))) \
           $(addprefix $(1)/rcg_dma.,$(addsuffix .o,$(DMA)))
OBJS    := $(call objs,$(OBJDIR))
# headers: every object depends on exactly the headers it includes (-MMD -MP writes a .d beside each .o; an object built
# before the .d files existed has none and falls back to "all headers")
ALLHDRS := $(wildcard $(CSRC)/*.hpp) $(ROOT)include/rcg.h
hdrs     = $(if $(wildcard $(1:.o=.d)),,$(ALLHDRS))
.SECONDEXPANSION:

DEVOBJDIR := $(ROOT)build/obj_dev
DEVOBJS   := $(call objs,$(DEVOBJDIR))
ASANDIR   := $(ROOT)build/asan
ASANOBJS  := $(call objs,$(ASANDIR))
SANFLAGS  := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1
HOSTCLANG ?= /opt/rocm/lib/llvm/bin/clang

all: lib oracle

lib: $(LIBDIR)/librcg.so
oracle: $(ORACLE)/_build/liboracle.so

$(OBJDIR)/rcg_dma.%.o: $(CSRC)/rcg_dma_inst.hip $$(call hdrs,$$@)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(DMAFLAGS) -c $< -o $@

$(OBJDIR)/rcg_sys.%.o: $(CSRC)/rcg_sys_inst.hip $$(call hdrs,$$@)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(SYSFLAGS) -c $< -o $@

$(OBJDIR)/%.o: $(CSRC)/%.hip $$(call hdrs,$$@)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/librcg.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--no-undefined $(OBJS) -o $@

$(ORACLE)/_build/liboracle.so: $(ORACLE)/oracle.c
	@mkdir -p $(ORACLE)/_build
	$(CC) -O2 -std=c11 -fPIC -shared -fopenmp -ffp-contract=off -Wall $(ORACLE)/oracle.c -o $@ -lm

dev: $(LIBDIR)/librcg_dev.so

$(DEVOBJDIR)/rcg_dma.%.o: $(CSRC)/rcg_dma_inst.hip $$(call hdrs,$$@)
	@mkdir -p $(DEVOBJDIR)
	$(HIPCC) $(HIPFLAGS) -DRCG_DEV $(DMAFLAGS) -c $< -o $@

$(DEVOBJDIR)/rcg_sys.%.o: $(CSRC)/rcg_sys_inst.hip $$(call hdrs,$$@)
	@mkdir -p $(DEVOBJDIR)
	$(HIPCC) $(HIPFLAGS) -DRCG_DEV $(SYSFLAGS) -c $< -o $@

$(DEVOBJDIR)/%.o: $(CSRC)/%.hip $$(call hdrs,$$@)
	@mkdir -p $(DEVOBJDIR)
	$(HIPCC) $(HIPFLAGS) -DRCG_DEV -c $< -o $@

$(LIBDIR)/librcg_dev.so: $(DEVOBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--no-undefined $(DEVOBJS) -o $@

# A/B build for tools/ab_lib.py: the same sources with extra defines (`make ab ABFLAGS="-DRCG_AB_..."`), linked as
# rcognita_amd/lib/librcg_ab.so; never loaded by the package (a tool binds it with _native.use_library)
ab: $(LIBDIR)/librcg_ab.so
ABOBJDIR := $(ROOT)build/obj_ab
ABOBJS   := $(call objs,$(ABOBJDIR))
$(ABOBJDIR)/rcg_dma.%.o: $(CSRC)/rcg_dma_inst.hip $$(call hdrs,$$@)
	@mkdir -p $(ABOBJDIR)
	$(HIPCC) $(HIPFLAGS) $(ABFLAGS) $(DMAFLAGS) -c $< -o $@

$(ABOBJDIR)/rcg_sys.%.o: $(CSRC)/rcg_sys_inst.hip $$(call hdrs,$$@)
	@mkdir -p $(ABOBJDIR)
	$(HIPCC) $(HIPFLAGS) $(ABFLAGS) $(SYSFLAGS) -c $< -o $@

$(ABOBJDIR)/%.o: $(CSRC)/%.hip $$(call hdrs,$$@)
	@mkdir -p $(ABOBJDIR)
	$(HIPCC) $(HIPFLAGS) $(ABFLAGS) -c $< -o $@

$(LIBDIR)/librcg_ab.so: $(ABOBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--no-undefined $(ABOBJS) -o $@

# Sanitizer build (CPU only: GPU AddressSanitizer is not available on this pool).  --offload-host-only compiles the
# host side of every .hip unit - the C ABI, argument checks, launch-geometry arithmetic - and drops the device code.
asan: $(ASANDIR)/abi_asan
ASANHIP := -std=c++17 --offload-arch=$(ARCH) --offload-host-only $(SANFLAGS) -fPIC -Wall -Wno-unused-function \
           -ffp-contract=fast -I$(ROOT)include

$(ASANDIR)/rcg_dma.%.o: $(CSRC)/rcg_dma_inst.hip $$(call hdrs,$$@)
	@mkdir -p $(ASANDIR)
	$(HIPCC) $(ASANHIP) $(DMAFLAGS) -c $< -o $@

$(ASANDIR)/rcg_sys.%.o: $(CSRC)/rcg_sys_inst.hip $$(call hdrs,$$@)
	@mkdir -p $(ASANDIR)
	$(HIPCC) $(ASANHIP) $(SYSFLAGS) -c $< -o $@

$(ASANDIR)/%.o: $(CSRC)/%.hip $$(call hdrs,$$@)
	@mkdir -p $(ASANDIR)
	$(HIPCC) $(ASANHIP) -c $< -o $@

$(ASANDIR)/asan_driver.o: $(ROOT)tests/asan_driver.c $(ORACLE)/oracle.c $(ROOT)include/rcg.h
	@mkdir -p $(ASANDIR)
	$(HOSTCLANG) -std=c11 $(SANFLAGS) -ffp-contract=off -Wall -I$(ROOT)include -c $< -o $@

# a host-only object still refers to the device image of its unit (__hip_fatbin_<hash>, registered with the runtime at
# load time, read only when a kernel is first launched): there is none in this build, so each gets an empty definition
$(ASANDIR)/no_device_image.c: $(ASANOBJS)
	nm -u $(ASANOBJS) | grep -o '__hip_fatbin_[0-9a-f]*' | sort -u | \
	  sed 's/.*/const unsigned char &[64] = {0};/' > $@

$(ASANDIR)/abi_asan: $(ASANOBJS) $(ASANDIR)/asan_driver.o $(ASANDIR)/no_device_image.c
	$(HOSTCLANG) -c $(ASANDIR)/no_device_image.c -o $(ASANDIR)/no_device_image.o
	$(HIPCC) $(SANFLAGS) $(ASANOBJS) $(ASANDIR)/asan_driver.o $(ASANDIR)/no_device_image.o -o $@ -lm

clean:
	rm -rf $(LIBDIR)/librcg.so $(LIBDIR)/librcg_dev.so $(LIBDIR)/librcg_ab.so $(OBJDIR) $(DEVOBJDIR) $(ABOBJDIR) $(ASANDIR) $(ORACLE)/_build

# (the included dependency files are not targets: without this rule make tries to REMAKE them through its built-in
# "link an executable from a .o" rule and compiles rcg_sys.*.d.o objects)
%.d: ;
-include $(wildcard $(OBJDIR)/*.d) $(wildcard $(DEVOBJDIR)/*.d) $(wildcard $(ABOBJDIR)/*.d)

.PHONY: all lib oracle dev ab asan clean
