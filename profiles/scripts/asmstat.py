"""Static instruction mix of one kernel in a hipcc -S listing: python asmstat.py file.s <mangled-substring> [--dump]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ": " in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
cnt = collections.Counter()
for l in body:
    t = l.strip()
    if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
        continue
    op = t.split()[0]
    cnt[op] += 1
groups = collections.Counter()
for op, n in cnt.items():
    if op.startswith("v_pk_fma"): g = "v_pk_fma"
    elif op.startswith("v_pk_"): g = "v_pk_other"
    elif op.startswith("v_mfma"): g = "mfma"
    elif "dpp" in op or op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")): g = "crosslane"
    elif op.startswith("v_"): g = "valu_other"
    elif op.startswith("ds_"): g = "lds"
    elif op.startswith(("global_", "buffer_", "flat_")): g = "vmem"
    elif op.startswith("s_waitcnt"): g = "s_waitcnt"
    elif op.startswith("s_nop"): g = "s_nop"
    elif op.startswith("s_"): g = "salu"
    else: g = "other"
    groups[g] += n
print(len(body), "lines;", sum(cnt.values()), "instructions")
print(dict(groups))
print(cnt.most_common(40))
if "--dump" in sys.argv:
    print("\n".join(body))
