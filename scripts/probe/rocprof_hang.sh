export TMPDIR=/tmp
cd /root/repo
run() { tag=$1; shift; timeout 60 rocprofv3 --kernel-trace --stats -d /tmp/r_$tag -- python3 scripts/probe/rocprof_hang.py "$@" > /tmp/r_$tag.log 2>&1; echo "$tag rc=$?"; grep -v "^W2026\|amdgpu.ids\|^E2026" /tmp/r_$tag.log | grep -v "^  File \"/usr" | tail -2 | cut -c1-160; }
run c2 c2 65536
run unf unfused 65536
