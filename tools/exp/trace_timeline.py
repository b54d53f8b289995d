"""trace_timeline.py <dir with *_kernel_trace.csv / *_memory_copy_trace.csv> -- kernels and bursts of copies of the last host-pointer call, in ms"""
import csv, glob, sys
d = sys.argv[1]
ks = list(csv.DictReader(open(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0])))
cs = list(csv.DictReader(open(glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True)[0])))
nk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kp = [k for k in ks if "k_pairing" in k["Kernel_Name"] or "k_layout" in k["Kernel_Name"] or "copyBuffer" in k["Kernel_Name"]]
last = [k for k in ks if "k_pairing" in k["Kernel_Name"]][-nk:]
t0 = int(last[0]["Start_Timestamp"])
ev = []
for k in kp:
    s = int(k["Start_Timestamp"]) - t0
    if s > -3e6:
        ev.append((s, int(k["End_Timestamp"]) - t0, "KERNEL " + k["Kernel_Name"].split("(")[0][-24:], k.get("Stream_Id"), 1))
b = []
for c in sorted(cs, key=lambda c: int(c["Start_Timestamp"])):
    s, e, dr, st = int(c["Start_Timestamp"]) - t0, int(c["End_Timestamp"]) - t0, c["Direction"][12:], c["Stream_Id"]
    if s < -3e6:
        continue
    if b and s - b[-1][1] < 0.1e6 and dr == b[-1][2] and st == b[-1][3]:
        b[-1][1] = e; b[-1][4] += 1
    else:
        b.append([s, e, dr, st, 1])
ev += [tuple(x) for x in b]
for s, e, n, st, cnt in sorted(ev)[:int(sys.argv[3]) if len(sys.argv) > 3 else 80]:
    print(f"{s/1e6:9.3f} {e/1e6:9.3f} {(e-s)/1e6:7.3f}  st{st} {n} x{cnt}")
