# Round-end measurement pass (one gpurun call): PMC passes, bench line, kernel traces, timing tools -> gpurun_out/r05p/
# (copy what is to be judged into profiles/ afterwards: tools/collect_profiles.sh).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
mkdir -p $O
# whatever happens, the raw rocprofv3 directories do not travel home (gpurun_out/ is merged back only below 64 MiB)
trap 'rm -rf $O/kt $O/kh $O/pmc/FETCH_SIZE $O/pmc/WRITE_SIZE $O/pmc/TCC* $O/pmc/SQ* $O/pmc_train/pass*/ $O/pmc_head*/pass*/ $O/pmc_va256/pass*/' EXIT
# the counter passes first: bench.py reads profiles/traffic.json, valu_insts.json and train_pmc.json and ignores them unless they
# carry the sha of the kernel sources it runs (they are copied into profiles/ here, on the box; copy them again from gpurun_out/ at home)
cd $R && bash tools/pmc_traffic.sh gpurun_out/r05p/pmc > $O/pmc.log 2>&1
cp $O/pmc/traffic.json $O/pmc/valu_insts.json $R/profiles/
echo pmc done
bash tools/pmc_sq.sh gpurun_out/r05p/pmc_train tools/prof_train_kernels.py 2 > $O/pmc_train.log 2>&1
python3 tools/pmc_dispatch_table.py gpurun_out/r05p/pmc_train train > $O/train_pmc_table.csv
python3 tools/pmc_train_json.py $O/train_pmc_table.csv 2 $O/train_pmc.json > /dev/null
cp $O/train_pmc.json $R/profiles/
echo pmc train done
for B in 10000 40960; do
  bash tools/pmc_sq.sh gpurun_out/r05p/pmc_head$B tools/prof_headline.py $B 5 > $O/pmc_head$B.log 2>&1
  python3 tools/pmc_dispatch_table.py gpurun_out/r05p/pmc_head$B dealt | (read h; echo "$h"; tail -5) > $O/headline_pmc_$B.csv
done
bash tools/pmc_sq.sh gpurun_out/r05p/pmc_va256 tools/prof_va256.py 125000 3 > $O/pmc_va256.log 2>&1
python3 tools/pmc_dispatch_table.py gpurun_out/r05p/pmc_va256 va256_wave | (read h; echo "$h"; tail -3) > $O/va256_pmc.csv
echo pmc headline + va256 done
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sustained > /dev/null 2>&1
python3 $R/tools/stats_by_grid.py $O/kt/t_kernel_trace.csv > $O/kernel_durations_by_grid.csv
cp $O/kt/t_kernel_stats.csv $O/kernel_stats.csv
echo kt done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kh -o t -- python3 $R/bench.py --no-cpu-baseline --no-configs --skip-fused-count --no-sustained > /dev/null 2>&1
cp $O/kh/t_kernel_stats.csv $O/headline_kernel_stats.csv
python3 $R/tools/stats_by_grid.py $O/kh/t_kernel_trace.csv > $O/headline_by_grid.csv
echo kh done
cd $R
python3 tools/prof_train_kernels.py 3 2>/dev/null > $O/train_kernels_time.csv
python3 tools/time_trials.py 1 6 16 51 102 204 256 512 2>&1 | grep -v amdgpu.ids > $O/time_trials.txt
python3 tools/time_step.py 2>&1 | grep -v amdgpu.ids > $O/time_step.txt
python3 tools/time_online.py 2>&1 | grep -v amdgpu.ids > $O/time_online_training.txt
python3 tools/time_online_states.py 2>&1 | grep -v amdgpu.ids > $O/time_online_states.txt
python3 tools/time_vnet_states.py 4,10000 8,10000 32,10000 64,10000 128,4000 128,10000 256,2000 4,1000 128,1000 2>&1 | grep -v amdgpu.ids > $O/time_vnet_states.txt
python3 tools/time_dealt.py 2>&1 | grep -v amdgpu.ids > $O/time_dealt.txt
python3 tools/time_survivors.py 2>&1 | grep -v amdgpu.ids > $O/time_survivors.txt
python3 tools/time_montecarlo.py 2>&1 | grep -v amdgpu.ids > $O/time_montecarlo.txt
timeout -k 10 200 python3 tools/fuzz_parity.py 120 2>&1 | grep -v amdgpu.ids | tail -5 > $O/fuzz_parity.txt
rm -rf $O/kt $O/kh $O/pmc/FETCH_SIZE $O/pmc/WRITE_SIZE $O/pmc/TCC* $O/pmc/SQ* $O/pmc_train/pass*/ $O/pmc_head*/pass*/ $O/pmc_va256/pass*/
ls $O $O/pmc
