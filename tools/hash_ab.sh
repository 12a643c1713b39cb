cd $GRAFT_REPO_ROOT
SMPLR_RASTER=1 python tools/probes/seg_hash.py > gpurun_out/hash_v1.txt 2>&1
SMPLR_RASTER=2 python tools/probes/seg_hash.py > gpurun_out/hash_v2.txt 2>&1
SMPLR_RASTER=2 SMPLR_RASTER_NG=4 python tools/probes/seg_hash.py > gpurun_out/hash_v2n4.txt 2>&1
diff gpurun_out/hash_v1.txt gpurun_out/hash_v2.txt && echo "v2 NG8: IDENTICAL"
diff gpurun_out/hash_v1.txt gpurun_out/hash_v2n4.txt && echo "v2 NG4: IDENTICAL"
tail -12 gpurun_out/hash_v2.txt
