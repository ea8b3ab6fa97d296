# Builds the gfx950 HIP engine in-tree:  bayesiannetwork_amd/libbn_hip.so
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := bayesiannetwork_amd/csrc
OUT   := bayesiannetwork_amd/libbn_hip.so
HDRS  := $(wildcard $(SRC)/*.h) include/bn_hip.h
FLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Iinclude

all: $(OUT)

$(OUT): $(SRC)/api.hip $(SRC)/plan.cpp $(HDRS)
	$(HIPCC) $(FLAGS) -shared -o $@ $(SRC)/api.hip $(SRC)/plan.cpp

clean:
	rm -f $(OUT)

.PHONY: all clean
