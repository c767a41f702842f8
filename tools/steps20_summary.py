import sys
rows=[l.split() for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/s20b/dump.txt")]
idx=[i for i,r in enumerate(rows) if len(r)>6 and "cursor_set" in r[6]]
i0=idx[-2]
for r in rows[max(0,i0-2):i0+12]: print(" ".join(r)[:110])
ad=[r for r in rows[i0:] if len(r)>6 and "adam" in r[6]]
first=float(rows[i0+1][1])
print("first kernel after cursor_set at", first, "last adam end", float(ad[19][1])+float(ad[19][3]), "span", float(ad[19][1])+float(ad[19][3])-first)
prev=None
for k in range(20):
  end=float(ad[k][1])+float(ad[k][3])
  print(k, ad[k][0], "adam dur", ad[k][3], "gap", ad[k][5], "step", round(end-(prev if prev else first),1))
  prev=end
