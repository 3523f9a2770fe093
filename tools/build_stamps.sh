#!/bin/sh
# diagnostic build of the C-ABI library with s_memtime phase stamps (never shipped, never benchmarked)
cd "$(dirname "$0")/../open_knowledge_graph_embeddings_amd/csrc" && \
hipcc -O3 --offload-arch=gfx950 -std=c++17 -DOKGE_STAMPS -shared -fPIC -o ../libokge_hip_stamps.so okge_api.hip okge_train.hip okge_train32.hip okge_train64.hip okge_train64k.hip okge_misc.hip okge_pool.hip okge_collate.cpp okge_dataset.cpp
