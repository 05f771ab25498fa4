# build a variant of the library with extra -D flags: build_variant.sh <suffix> <flags...>  -> surge_amd/libsurge_replay_<suffix>.so
suf=$1; shift
cd "$(dirname "$0")/../.."
S=surge_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude "$@" \
  $S/fold_kernels.hip $S/fold_chunked.hip $S/fold_tiled.hip $S/fold_slots.hip $S/state_kernels.hip $S/stream_kernels.hip $S/engine.hip $S/comm.hip \
  $S/ingest.cpp $S/event_decode.cpp $S/lz4_frame.cpp $S/snapshot_writer.cpp -o surge_amd/libsurge_replay_$suf.so
