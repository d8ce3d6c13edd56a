import subprocess, os, sys
D = "/dev/shm/e2e8m"
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
segs = ",".join("%s/left_seg%d.bam" % (D, k) for k in (1, 2, 3, 4))
cmd = [ROOT + "/tophat_amd/bin/long_spanning_reads", "--segment-length", "25", "--sam-header", D + "/hdr.sam", D + "/ref.fa", D + "/left_reads.bam",
       D + "/o.j", D + "/o.i", D + "/o.d", "/dev/null", D + "/x.bam", segs]
for i in range(8):
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, THJ_TIMING="1", THJ_BGZF_LEVEL="1"))
    print(i, "rc", r.returncode, "stderr tail:", r.stderr.strip().splitlines()[-2:], flush=True)
    if r.returncode != 0:
        print(r.stderr[-1500:])
