import re,collections,sys
rows=[l for l in open(sys.argv[1]) if l.startswith("q")]
tot=collections.defaultdict(float); cnt=collections.Counter()
for l in rows:
    m=re.match(r"q(\d+)\s+(\S+.*?)\s+grid\s+\S+\s+wg\s+\S+\s+start\s+([\d.]+) us\s+dur\s+([\d.]+) us",l)
    if m: tot[(m.group(1),m.group(2).strip())]+=float(m.group(4)); cnt[(m.group(1),m.group(2).strip())]+=1
for k,v in sorted(tot.items(), key=lambda kv:-kv[1])[:int(sys.argv[2])]: print(k, cnt[k], round(v,1))
print(open(sys.argv[1]).read().strip().splitlines()[-1])
