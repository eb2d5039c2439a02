export TMPDIR=/tmp
out=gpurun_out/prof_r06_issue_c3_f32; rm -rf $out; mkdir -p $out
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $out/b -- python3 bench.py --pmc-child --pmc-dtype f32 > $out/b.log 2>&1
f=$(find $out/b -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/b.db; rm -rf $out/b
python3 scripts/pmc_dump.py $out/b.db > $out/pmc.txt 2>&1
grep -A12 "cond_gf_split" $out/pmc.txt | head -40
