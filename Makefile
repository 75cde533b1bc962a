# Build the MI355X-native library (gfx950 only) and the CPU oracle.
#   make            -> rcognita_amd/lib/librcg.so  +  oracle/_build/liboracle.so
#   make lib        -> HIP library only (hipcc cross-compiles without a GPU)
#   make oracle     -> C oracle only (gcc)
HIPCC   ?= hipcc
CC      ?= gcc
ARCH    ?= gfx950
ROOT    := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
CSRC    := $(ROOT)rcognita_amd/csrc
LIBDIR  := $(ROOT)rcognita_amd/lib
ORACLE  := $(ROOT)oracle

HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Wall -Wno-unused-function \
            -ffp-contract=fast -I$(ROOT)include

all: lib oracle

lib: $(LIBDIR)/librcg.so
oracle: $(ORACLE)/_build/liboracle.so

$(LIBDIR)/librcg.so: $(CSRC)/rcg_api.hip $(CSRC)/rcg_kernels.hpp $(CSRC)/rcg_systems.hpp $(CSRC)/rcg_math.hpp $(ROOT)include/rcg.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(CSRC)/rcg_api.hip -o $@

$(ORACLE)/_build/liboracle.so: $(ORACLE)/oracle.c
	@mkdir -p $(ORACLE)/_build
	$(CC) -O2 -std=c11 -fPIC -shared -fopenmp -ffp-contract=off -Wall $(ORACLE)/oracle.c -o $@ -lm

clean:
	rm -rf $(LIBDIR)/librcg.so $(ORACLE)/_build

.PHONY: all lib oracle clean
