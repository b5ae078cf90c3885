# Builds libdann_hip.so (the C-ABI shared library of include/dann.h) for gfx950 without Python:
#   make -j8            -> diskann_amd/libdann_hip.so
#   make example        -> examples/c_api_example (plain C caller)
# `python -m diskann_amd.build` does the same with dependency tracking; __graft_entry__.build() uses that.
HIPCC   ?= $(shell command -v hipcc 2>/dev/null || echo /opt/rocm/bin/hipcc)
ARCH    ?= gfx950
FLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero
CSRC    := diskann_amd/csrc
OBJDIR  := diskann_amd/build
SOURCES := api search_kernels search_f32 search_f16 search_u8 search_i8 search_sq8 search_pq search_pqlut search_pqlut2 search_pqlut3 search_pqlut4 search_pair server sharded paged_kernels \
           distance_kernels build_kernels pq_kernels
OBJS    := $(SOURCES:%=$(OBJDIR)/%.o)
HEADERS := $(wildcard $(CSRC)/*.h) include/dann.h
LIB     := diskann_amd/libdann_hip.so

all: $(LIB)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HEADERS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

example: $(LIB)
	gcc -std=c99 -O2 -Wall -Wextra -Iinclude examples/c_api_example.c -Ldiskann_amd -ldann_hip \
	    -Wl,-rpath,$(abspath diskann_amd) -lm -o examples/c_api_example

clean:
	rm -rf $(OBJDIR) $(LIB) examples/c_api_example

.PHONY: all example clean
