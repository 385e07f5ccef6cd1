cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_rs
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rs -- python /root/repo/bench.py --feed resrgan --netd unet --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variant > /dev/null 2>&1
f=$(find /tmp/prof_rs -name "*kernel_stats.csv" | head -1)
python - <<P
import csv
rows=list(csv.DictReader(open("$f")))
keys=("filter2d","resize","noise","jpeg","feed_u8","clamp","degrade","sinc","blur")
tot=0
out=[]
for r in rows:
    n=r["Name"]
    if any(k in n.lower() for k in keys):
        ms=float(r["TotalDurationNs"])/1e6/4
        tot+=ms
        out.append("%-70s %6d calls/step %8.3f ms/step" % (n[:70], int(r["Calls"])//4, ms))
open("/root/repo/gpurun_out/r03ak_resrgan_feed_kernels.txt","w").write("\n".join(out)+"\ntotal %.3f ms/step\n" % tot)
print("\n".join(out)); print("total", tot)
P
