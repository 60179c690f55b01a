# Run ON THE GPU BOX: the round's profile set.   bash tools/final_profiles.sh <tag>      (e.g. r04)
# -> gpurun_out/<tag>_results/ (one bench line per single-GPU BASELINE config), gpurun_out/prof_<tag>_*/
#    (rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes), instruction mixes; tools/profile_commit.py
#    copies what is to be tracked into profiles/.
set -u
tag=${1:?tag}
cd $GRAFT_REPO_ROOT
bash tools/run_configs.sh gpurun_out/${tag}_results > gpurun_out/${tag}_configs.log 2>&1
bash tools/profile_gpu.sh ${tag}_text_p16 --workload text > gpurun_out/${tag}_prof_text.log 2>&1
bash tools/profile_gpu.sh ${tag}_low_p16 --workload low > gpurun_out/${tag}_prof_low.log 2>&1
bash tools/profile_gpu.sh ${tag}_urls_p16 --workload urls > gpurun_out/${tag}_prof_urls.log 2>&1
bash tools/profile_gpu.sh ${tag}_page_p13 --workload page > gpurun_out/${tag}_prof_page.log 2>&1
bash tools/profile_gpu.sh ${tag}_text_p15 --workload text --p 15 > gpurun_out/${tag}_prof_text15.log 2>&1
bash tools/pmc_insts.sh ${tag}_text_p16 --workload text > gpurun_out/${tag}_insts_text.log 2>&1
bash tools/pmc_insts.sh ${tag}_low_p16 --workload low > gpurun_out/${tag}_insts_low.log 2>&1
cat gpurun_out/${tag}_configs.log
