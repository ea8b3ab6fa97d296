# Builds the gfx950 HIP engine in-tree:  bayesiannetwork_amd/libbn_hip.so
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := bayesiannetwork_amd/csrc
OUT   := bayesiannetwork_amd/libbn_hip.so
OBJ   := build
# -MMD: every object depends on exactly the headers it includes (build/*.d)
FLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Iinclude -MMD -MP
OBJS  := $(OBJ)/api.o $(OBJ)/leaf_f32.o $(OBJ)/leaf_f64.o $(OBJ)/exact.o $(OBJ)/sampler.o $(OBJ)/plan.o $(OBJ)/packs.o
# Cardinality packs built with the library (csrc/cardinality_pack.h): the fused kernels for wide parents
# with these numbers of states.  One more:  make pack D0=14
PACKS ?= 6 7 8 9 11 12 13 15 16 20 24 32
PACK_OUT = bayesiannetwork_amd/libbn_hip_d$(1).so
# ... and shape packs: the fused kernel of leaf families with two wide parents for these CPT shapes
# [D0]x[D1]x[S].  One more:  make pack2 SHAPE=2x5x2
PACKS2 ?= 3x3x4 2x4x2 4x2x2 2x2x5
PACK2_OUT = bayesiannetwork_amd/libbn_hip_p$(1).so

all: $(OUT) packs

packs: $(foreach d,$(PACKS),$(call PACK_OUT,$(d))) $(foreach x,$(PACKS2),$(call PACK2_OUT,$(x)))

pack2: $(call PACK2_OUT,$(SHAPE))

# (SHAPE = D0xD1xS)
bayesiannetwork_amd/libbn_hip_p%.so: $(SRC)/pack2.hip $(wildcard $(SRC)/*.h)
	$(HIPCC) -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -DBN_PACK2_D0=$(word 1,$(subst x, ,$*)) -DBN_PACK2_D1=$(word 2,$(subst x, ,$*)) -DBN_PACK2_S=$(word 3,$(subst x, ,$*)) -shared -Wl,-z,defs -o $@ $<

pack: $(call PACK_OUT,$(D0))

bayesiannetwork_amd/libbn_hip_d%.so: $(SRC)/pack.hip $(wildcard $(SRC)/*.h)
	$(HIPCC) -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -DBN_PACK_D0=$* -shared -Wl,-z,defs -o $@ $<

$(OUT): $(OBJS)
	$(HIPCC) -fPIC --offload-arch=$(ARCH) -shared -o $@ $(OBJS)

$(OBJ)/%.o: $(SRC)/%.hip
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c -o $@ $<

$(OBJ)/%.o: $(SRC)/%.cpp
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c -o $@ $<

# plain C++ against the C ABI (no Python, no torch): examples/c_abi_demo
demo: examples/c_abi_demo

examples/c_abi_demo: examples/c_abi_demo.cpp $(OUT) include/bn_hip.h
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -Iinclude -o $@ $< -L$(dir $(OUT)) -lbn_hip -Wl,-rpath,'$$ORIGIN/../$(dir $(OUT))'

-include $(OBJS:.o=.d)

clean:
	rm -rf $(OUT) $(OBJ) examples/c_abi_demo bayesiannetwork_amd/libbn_hip_d*.so bayesiannetwork_amd/libbn_hip_p*.so

.PHONY: all clean demo packs pack pack2
