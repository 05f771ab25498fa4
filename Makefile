# Plain-make build for hosts that do not go through Python (the JNI shim, the C / C++ demos).
# `python -c "import __graft_entry__ as g; g.build()"` runs the same hipcc command.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
LIB     := surge_amd/libsurge_replay.so
SRC     := surge_amd/csrc/fold_kernels.hip surge_amd/csrc/fold_chunked.hip surge_amd/csrc/fold_tiled.hip surge_amd/csrc/fold_slots.hip surge_amd/csrc/index_kernels.hip surge_amd/csrc/ingest_kernels.hip surge_amd/csrc/rtc.cpp surge_amd/csrc/f64_text.cpp surge_amd/csrc/state_kernels.hip surge_amd/csrc/stream_kernels.hip surge_amd/csrc/engine.hip surge_amd/csrc/comm.hip surge_amd/csrc/ingest.cpp surge_amd/csrc/event_decode.cpp surge_amd/csrc/lz4_frame.cpp surge_amd/csrc/snapshot_writer.cpp
HDR     := include/surge_replay.h include/surge_ingest.h include/surge_snapshot.h surge_amd/csrc/replay_internal.h surge_amd/csrc/fold_layout.h surge_amd/csrc/fold_device.h surge_amd/csrc/fold_chunk_device.h surge_amd/csrc/fold_slots_device.h surge_amd/csrc/f64_text.h surge_amd/csrc/f64_parse.h
LDDEMO  := -Lsurge_amd -lsurge_replay -Wl,-rpath,$(CURDIR)/surge_amd -L/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib

.PHONY: all lib oracle demos clean
all: lib oracle

lib: $(LIB)
$(LIB): $(SRC) $(HDR)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Iinclude -Isurge_amd/csrc $(SRC) -o $@

oracle:
	$(MAKE) -C oracle

demos: lib
	gcc -std=c99 -Wall -Iinclude examples/c_host_demo.c $(LDDEMO) -o examples/c_host_demo
	g++ -std=c++17 -Wall -Iinclude examples/cpp_host_demo.cpp $(LDDEMO) -o examples/cpp_host_demo

clean:
	rm -f $(LIB) examples/c_host_demo examples/cpp_host_demo
	$(MAKE) -C oracle clean
