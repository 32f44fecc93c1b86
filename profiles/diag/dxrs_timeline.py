import sys, collections
rows=[l.split() for l in sys.stdin if l.startswith("DXRSALL")]
rows=[(int(r[1]),int(r[2]),int(r[3]),int(r[4]),int(r[5]),int(r[6])) for r in rows]
# group launches by start-time clusters: sort by pe0, split when gap > 2000 ticks
rows.sort(key=lambda r:r[3])
launches=[];cur=[rows[0]]
for r in rows[1:]:
    if r[3]-cur[-1][3] > 5000: launches.append(cur); cur=[]
    cur.append(r)
launches.append(cur)
print(len(launches),'launches', [len(l) for l in launches][:8])
L=launches[len(launches)//2]
t0=min(r[3] for r in L)
starts=sorted(r[3]-t0 for r in L); ends=sorted(r[5]-t0 for r in L); durs=sorted(r[5]-r[3] for r in L)
print('start ticks (10ns): min %d med %d max %d'%(starts[0],starts[len(starts)//2],starts[-1]))
print('end: min %d med %d max %d'%(ends[0],ends[len(ends)//2],ends[-1]))
print('wg duration: min %d med %d max %d'%(durs[0],durs[len(durs)//2],durs[-1]))
# CU occupancy: hwid bits: cu_id [11:8], sh_id 12, se_id [15:13]? print distinct (se,sh,cu) count and max wgs per cu
def cu(h): return (h>>8)&0xFF | ((h>>16)&0xF)<<8
c=collections.Counter((r[2]>>8)&0xFFFF for r in L)
print('distinct cu ids', len(c), 'max wgs on one', max(c.values()), collections.Counter(c.values()))
