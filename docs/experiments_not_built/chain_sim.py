import numpy as np, sys
def sim(nsimd=32, wps=4, chain=25, lag=6, length=1200, nchains=40, seed=1):
    rng=np.random.default_rng(seed)
    total=nchains*chain
    pos=np.zeros(total,int); started=np.zeros(total,bool); done=np.zeros(total,bool)
    slots=[[] for _ in range(nsimd)]   # wave ids per simd
    next_ticket=0
    # initial fill: hardware places workgroups round-robin over simds (ticket order = arrival order)
    order=[k for _ in range(wps) for k in range(nsimd)]
    for k in order:
        if next_ticket<total:
            slots[k].append(next_ticket); started[next_ticket]=True; next_ticket+=1
    rr=[0]*nsimd
    t=0; work=0
    ndone=0
    while ndone<total:
        t+=1
        for k in range(nsimd):
            ws=slots[k]; n=len(ws)
            for off in range(n):
                idx=(rr[k]+off)%n
                w=ws[idx]
                s=w%chain
                if s==0 or done[w-1] or pos[w-1]>=pos[w]+lag:
                    pos[w]+=1; work+=1
                    rr[k]=(idx+1)%n
                    if pos[w]>=length:
                        done[w]=True; ndone+=1
                        if next_ticket<total:
                            ws[idx]=next_ticket; started[next_ticket]=True; next_ticket+=1
                        else:
                            ws.pop(idx); 
                            if ws: rr[k]%=len(ws)
                            else: rr[k]=0
                    break
    return t, work/nsimd, work/nsimd/t
for wps in (2,4,5,8):
    print('wps',wps, sim(wps=wps))
print('lag 2', sim(lag=2)); print('lag 12', sim(lag=12))
print('chain 5', sim(chain=5,nchains=200)); print('chain 50', sim(chain=50,nchains=20))
