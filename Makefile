# Build libdeepspeaker_hip.so (gfx950) in-tree.  `python -c "import __graft_entry__ as g; g.build()"`
# drives the same recipe.
HIPCC ?= /opt/rocm/bin/hipcc
PKG   := deepspeaker-pytorch_amd
CSRC  := $(PKG)/csrc
SRCS  := $(wildcard $(CSRC)/*.hip)
HDRS  := $(wildcard $(CSRC)/*.h) include/deepspeaker_hip.h
LIB   := $(PKG)/libdeepspeaker_hip.so
FLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$(CSRC) -Iinclude -Wall -Wno-unused-function -Wno-pass-failed

all: $(LIB)

$(LIB): $(SRCS) $(HDRS)
	$(HIPCC) $(FLAGS) -o $@ $(SRCS)

clean:
	rm -f $(LIB)
.PHONY: all clean
