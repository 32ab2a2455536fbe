import sys,glob,re
for path in sorted(glob.glob(sys.argv[1]+'/tl_*.txt')):
    m=re.search(r'H(\d+)_Ci(\d+)_Co(\d+)',path); H,Ci,Co=map(int,m.groups())
    nch=Ci//16
    for ln in open(path):
        if not ln.startswith('wg0 wave4'): continue
        ev=[tuple(map(int,t.split(':'))) for t in ln.split(':',1)[1].split()]
        # item boundaries: marks 4 (after switch) ; first item starts at ev[0]
        starts=[ev[0][1]]+[t for k,t in ev if k==4]
        ends=[t for k,t in ev if k==3]
        per=[b-a for a,b in zip(starts,starts[1:])]
        epi=[]
        t2=None
        for k,t in ev:
            if k==2: t2=t
            if k==4 and t2: epi.append(t-t2); t2=None
        chunk=[]
        prev=None
        for k,t in ev:
            if k==1:
                if prev is not None and lastk!=4: chunk.append(t-prev)
                prev=t
            lastk=k if k in (1,4) else (lastk if 'lastk' in dir() else None)
        n=len(per)
        if n==0: print(path, 'single item'); continue
        avg=sum(per)/n
        print("H%-3d Ci%-4d Co%-4d items(seen)=%2d period=%7.0f ideal=%7.0f (%.0f%%)  epi+switch=%6.0f (%.0f%%)  per-chunk=%.0f" % (H,Ci,Co,n,avg,nch*6912,100*nch*6912/avg, sum(epi)/max(len(epi),1), 100*sum(epi)/max(len(epi),1)/avg, (avg-sum(epi)/max(len(epi),1))/nch))
