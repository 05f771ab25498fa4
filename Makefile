# Plain-make build for hosts that do not go through Python (the JNI shim, the C / C++ demos).
# `python -c "import __graft_entry__ as g; g.build()"` runs the same hipcc command over the same list of sources
# (surge_amd/csrc/SOURCES; tests/test_abi.py builds through this Makefile and checks the exports).
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
LIB     ?= surge_amd/libsurge_replay.so
OBJ     ?= build/make
FLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall
SRC     := $(addprefix surge_amd/csrc/,$(shell grep -v "^\#" surge_amd/csrc/SOURCES))
HDR     := include/surge_replay.h include/surge_ingest.h include/surge_snapshot.h surge_amd/csrc/replay_internal.h surge_amd/csrc/fold_layout.h surge_amd/csrc/fold_device.h surge_amd/csrc/fold_chunk_device.h surge_amd/csrc/fold_lane_device.h surge_amd/csrc/fold_flat_device.h surge_amd/csrc/fold_slots_device.h surge_amd/csrc/f64_text.h surge_amd/csrc/f64_parse.h
LDDEMO  := -Lsurge_amd -lsurge_replay -Wl,-rpath,$(CURDIR)/surge_amd -L/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib

.PHONY: all lib oracle demos clean
all: lib oracle

lib: $(LIB)
$(OBJ)/%.o: surge_amd/csrc/% $(HDR)
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -Iinclude -Isurge_amd/csrc -c $< -o $@
$(LIB): $(addprefix $(OBJ)/,$(addsuffix .o,$(notdir $(SRC))))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $^ -o $@

oracle:
	$(MAKE) -C oracle

demos: lib
	gcc -std=c99 -Wall -Iinclude examples/c_host_demo.c $(LDDEMO) -o examples/c_host_demo
	g++ -std=c++17 -Wall -Iinclude examples/cpp_host_demo.cpp $(LDDEMO) -o examples/cpp_host_demo

clean:
	rm -rf $(OBJ) $(LIB) examples/c_host_demo examples/cpp_host_demo
	$(MAKE) -C oracle clean
