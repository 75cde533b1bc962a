# Build the MI355X-native library (gfx950 only) and the CPU oracle.
#   make -j4        -> rcognita_amd/lib/librcg.so  +  oracle/_build/liboracle.so
#   make lib        -> HIP library only (hipcc cross-compiles without a GPU)
#   make oracle     -> C oracle only (gcc)
HIPCC   ?= hipcc
CC      := gcc
ARCH    ?= gfx950
ROOT    := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
CSRC    := $(ROOT)rcognita_amd/csrc
LIBDIR  := $(ROOT)rcognita_amd/lib
OBJDIR  := $(ROOT)build/obj
ORACLE  := $(ROOT)oracle

HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function -ffp-contract=fast \
            -I$(ROOT)include
UNITS   := rcg_api rcg_sys_3wrobot rcg_sys_3wrobotni rcg_sys_2tank
OBJS    := $(addprefix $(OBJDIR)/,$(addsuffix .o,$(UNITS)))
HDRS    := $(wildcard $(CSRC)/*.hpp) $(ROOT)include/rcg.h

all: lib oracle

lib: $(LIBDIR)/librcg.so
oracle: $(ORACLE)/_build/liboracle.so

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/librcg.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

$(ORACLE)/_build/liboracle.so: $(ORACLE)/oracle.c
	@mkdir -p $(ORACLE)/_build
	$(CC) -O2 -std=c11 -fPIC -shared -fopenmp -ffp-contract=off -Wall $(ORACLE)/oracle.c -o $@ -lm

clean:
	rm -rf $(LIBDIR)/librcg.so $(OBJDIR) $(ORACLE)/_build

.PHONY: all lib oracle clean
