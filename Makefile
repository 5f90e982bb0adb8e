# Build libdeepspeaker_hip.so (gfx950) in-tree: one object per kernel source (make -j compiles them in
# parallel), then one link.  `python -c "import __graft_entry__ as g; g.build()"` drives the same recipe.
HIPCC ?= /opt/rocm/bin/hipcc
PKG   := deepspeaker-pytorch_amd
CSRC  := $(PKG)/csrc
OBJD  := build/obj
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJD)/%.o,$(SRCS))
HDRS  := $(wildcard $(CSRC)/*.h) include/deepspeaker_hip.h
LIB   := $(PKG)/libdeepspeaker_hip.so
EXTRA ?=
FLAGS := $(EXTRA) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$(CSRC) -Iinclude -Wall -Wno-unused-function -Wno-pass-failed

all: $(LIB)

# the fp16 convolution's chunk body is one fully unrolled stream of up to 500 MFMAs with a side operation after
# each; above the default size limit `#pragma unroll` silently stops unrolling, and the register arrays indexed by
# the loop counter (filter ring, fragment buffers) would then live in scratch memory
$(OBJD)/conv_mfma_f16_k%.o $(OBJD)/conv_mfma_f16_pk%.o $(OBJD)/conv_block_f16.o: FLAGS += -mllvm -pragma-unroll-threshold=1000000

$(OBJD)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJD)
	$(HIPCC) $(FLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf $(LIB) $(OBJD)
.PHONY: all clean
