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
FLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$(CSRC) -Iinclude -Wall -Wno-unused-function -Wno-pass-failed

all: $(LIB)

$(OBJD)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJD)
	$(HIPCC) $(FLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf $(LIB) $(OBJD)
.PHONY: all clean
