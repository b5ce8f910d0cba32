for c in 22 24 26 28; do echo "chunk 2^$c"; LSDSORT_FEED_CHUNK_LOG2=$c python tools/host_entry_perf.py 2>&1 | grep "n=2^28"; done
