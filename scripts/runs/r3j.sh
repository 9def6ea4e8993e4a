#!/bin/bash
# round 3, GPU call J: rocprofv3 passes of every workload at HEAD (per phase) + the bench lines that go to profiles/
O=gpurun_out/r3j; mkdir -p $O
PREDICT=1 timeout 600 bash scripts/prof_workload.sh din > $O/p_din.log 2>&1
PREDICT=1 timeout 900 bash scripts/prof_workload.sh youtube --workload youtube > $O/p_youtube.log 2>&1
timeout 400 bash scripts/prof_workload.sh mlp --workload mlp > $O/p_mlp.log 2>&1
timeout 600 bash scripts/prof_workload.sh item2vec --workload item2vec > $O/p_item2vec.log 2>&1
timeout 400 bash scripts/prof_workload.sh knn --workload knn > $O/p_knn.log 2>&1
timeout 600 bash scripts/prof_workload.sh dinemb --train-emb 0.05 > $O/p_dinemb.log 2>&1
timeout 900 bash scripts/prof_workload.sh youtubeemb --workload youtube --train-emb 0.05 > $O/p_youtubeemb.log 2>&1
du -sh gpurun_out/p_*
