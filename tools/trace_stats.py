import numpy as np, sys
a=np.fromfile(sys.argv[1],dtype=np.uint64).reshape(-1,4)
a=a[a[:,0]>0]
t0=a[:,0].min()
freq=100e6  # wall_clock64: 100 MHz
st=(a[:,0]-t0)/freq*1e6; lp=(a[:,1]-t0)/freq*1e6; en=(a[:,2]-t0)/freq*1e6
rd=(a[:,3]&0xffffffff).astype(int)
print("waves",len(a))
for name,v in (("start",st),("loop",lp),("end",en)):
    print(name,"min %.1f p10 %.1f med %.1f p90 %.1f max %.1f us"%(v.min(),np.percentile(v,10),np.median(v),np.percentile(v,90),v.max()))
print("rounds min/med/max",rd.min(),np.median(rd),rd.max(), "sum",rd.sum())
print("setup (loop-start) med %.1f max %.1f"%(np.median(lp-st),(lp-st).max()))
print("per-round time med %.2f us"%np.median((en-lp)/np.maximum(rd,1)))
h,_=np.histogram(en,bins=10); print("end hist",h, "range %.0f..%.0f"%(en.min(),en.max()))
h,_=np.histogram(st,bins=10); print("start hist",h)
sm=(a[:,3]>>np.uint64(32)).astype(int)
print("distinct smid",len(set(sm)))
import collections
cnt=collections.Counter(sm)
print("waves per smid: ",collections.Counter(cnt.values()))
# rounds and end by smid
for k in sorted(cnt)[:12]:
    m=sm==k
    print("smid %4d waves %3d rounds %4d end max %.0f min %.0f"%(k,m.sum(),rd[m].sum(),en[m].max(),en[m].min()))
tot=np.array([rd[sm==k].sum() for k in sorted(cnt)]); print("rounds per smid: min %d med %d max %d"%(tot.min(),np.median(tot),tot.max()))
# order of start vs rounds
o=np.argsort(st); print("rounds of earliest 8 waves",rd[o[:8]],"latest 8",rd[o[-8:]])
print("corr(start, rounds) %.2f"%np.corrcoef(st,rd)[0,1])
m=sm==sorted(cnt)[0]
o=np.argsort(en[m])
print("one CU: (loop_start, end, rounds)")
print(np.c_[lp[m][o].round(0),en[m][o].round(0),rd[m][o]])
