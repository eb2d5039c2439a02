export TMPDIR=/tmp
out=gpurun_out/pmc_gb
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d $out/sq1 -- python3 scripts/probe/gb_bcast.py > $out/sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA -d $out/sq2 -- python3 scripts/probe/gb_bcast.py > $out/sq2.log 2>&1
for d in sq1 sq2; do f=$(find $out/$d -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/$d.db; rm -rf $out/$d; done
python3 scripts/pmc_dump.py $out/sq1.db $out/sq2.db 2>&1 | grep -i "bwd"
