"""Design check of csrc/fft256.hpp (no GPU, numpy only):  python tools/fft256_model.py

1. The index algebra of the half transforms — lane / register digits, the three digit swaps, every twiddle — restated in
   numpy with exact complex exponentials and compared with the direct O(n^2) definition of the folded negacyclic transform
   (forward: F_0 +- W^k' F_1 = A[k]; inverse: even / odd coefficients).
2. The LDS slot functions (low / mid / top swap) under the lane grouping of MI355X_MICROARCH.md: ds_write_b128 is serviced
   in 8 groups of 8 contiguous lanes (bank = slot mod 8), ds_read_b128 in the four 16-lane groups listed there (bank =
   slot mod 16): every access of both directions must be 1-way (conflict-free).
The C++ itself is checked against fft512.hpp by tests/test_kernel_emulation.py::test_half_transforms_equal_the_full_transform."""
import numpy as np, itertools
rng=np.random.default_rng(1)
M=512
W=np.exp(2j*np.pi/512); psi=np.exp(1j*np.pi/1024)
def d4(x,inv):  # x: list of 4 ; X[t]=sum x[s] (±i)^{st}
    w = -1j if inv else 1j
    return [sum(x[s]*w**(s*t) for s in range(4)) for t in range(4)]
# slots
def s_low(T,Mm,L,R): return 64*T+16*L+4*Mm+(R^L)
def s_mid(T,Mm,L,R): return 64*T+16*Mm+4*(R^Mm)+L
def s_top(T,Mm,L,R): return 64*R+16*T+4*Mm+L
def lane(a,b,c): return 16*a+4*b+c
def dig(l): return (l>>4)&3,(l>>2)&3,l&3

def inverse_half(C,p):
    """C: spectrum indexed by frequency k (512). returns dict j->z[j] for j=2m+p"""
    # stage in: lane (r0,r1,r2) reads C[r+64q]
    regs=np.zeros((64,4),complex)
    for l in range(64):
        r0,r1,r2=dig(l); r=r0+4*r1+16*r2
        c=[C[r+64*q] for q in range(8)]
        if p==0: y=[c[a]+c[a+4] for a in range(4)]
        else: y=[(c[a]-c[a+4])*np.exp(-2j*np.pi*a/8) for a in range(4)]
        Y=d4(y,True)
        for m0 in range(4): regs[l,m0]=Y[m0]*W**(-(2*m0+p)*r)
    # ex1 low swap: writer lane (T=r0,M=r1,L=r2) reg R=m0 -> reader lane (r0,r1,m0) reg r2
    xb=np.zeros(256,complex)
    for l in range(64):
        T,Mm,L=dig(l)
        for R in range(4): xb[s_low(T,Mm,L,R)]=regs[l,R]
    regs2=np.zeros((64,4),complex)
    for l in range(64):
        T,Mm,R=dig(l)
        for L in range(4): regs2[l,L]=xb[s_low(T,Mm,L,R)]
    # pass B: lane (r0,r1,m0) reg r2 -> n2 ; twiddle W^{-8 n2 (r0+4r1)}
    for l in range(64):
        r0,r1,m0=dig(l)
        Y=d4(list(regs2[l]),True)
        for n2 in range(4): regs2[l,n2]=Y[n2]*W**(-8*n2*(r0+4*r1))
    # ex2 mid swap: writer lane (T=r0,M=r1,L=m0) reg R=n2 -> reader lane (r0,n2,m0) reg r1
    for l in range(64):
        T,Mm,L=dig(l)
        for R in range(4): xb[s_mid(T,Mm,L,R)]=regs2[l,R]
    regs3=np.zeros((64,4),complex)
    for l in range(64):
        T,R,L=dig(l)
        for Mm in range(4): regs3[l,Mm]=xb[s_mid(T,Mm,L,R)]
    # pass C: lane (r0,n2,m0) reg r1 -> n1; twiddle W^{-32 n1 r0} * psi^{-(2 lam + p)}, lam=16n1+4n2+m0
    for l in range(64):
        r0,n2,m0=dig(l)
        Y=d4(list(regs3[l]),True)
        for n1 in range(4):
            lam=16*n1+4*n2+m0
            regs3[l,n1]=Y[n1]*W**(-32*n1*r0)*psi**(-(2*lam+p))
    # ex3 top swap: writer lane (T=r0,M=n2,L=m0) reg R=n1 -> reader lane (n1,n2,m0) reg r0
    for l in range(64):
        T,Mm,L=dig(l)
        for R in range(4): xb[s_top(T,Mm,L,R)]=regs3[l,R]
    out={}
    for l in range(64):
        R,Mm,L=dig(l)
        x=[xb[s_top(T,Mm,L,R)] for T in range(4)]
        Y=d4(x,True)
        for n0 in range(4):
            j=2*l+p+128*n0
            out[j]=Y[n0]*psi**(-128*n0)
    return out

def forward_half(z,p):
    """z: folded complex input z[j] j<512 (untwisted). returns F'[k'] (k'<256): p=0: F_0 ; p=1: W^{k'} F_1"""
    regs=np.zeros((64,4),complex)
    for l in range(64):
        x=[z[2*l+p+128*n0]*psi**(128*n0) for n0 in range(4)]
        Y=d4(x,False)
        for r0 in range(4): regs[l,r0]=Y[r0]*psi**(2*l+p)*W**(2*l*r0)
    xb=np.zeros(256,complex)
    # ex3': writer lane (n1,n2,m0) reg r0 -> reader lane (r0,n2,m0) reg n1   [slot s_top(T=r0,M=n2,L=m0,R=n1)]
    for l in range(64):
        R,Mm,L=dig(l)
        for T in range(4): xb[s_top(T,Mm,L,R)]=regs[l,T]
    regs2=np.zeros((64,4),complex)
    for l in range(64):
        T,Mm,L=dig(l)
        for R in range(4): regs2[l,R]=xb[s_top(T,Mm,L,R)]
    # pass 2: lane (r0,n2,m0) reg n1 -> r1 ; twiddle W64^{m'' r1}, m''=m0+4n2
    for l in range(64):
        r0,n2,m0=dig(l)
        Y=d4(list(regs2[l]),False)
        for r1 in range(4): regs2[l,r1]=Y[r1]*W**(8*(m0+4*n2)*r1)
    # ex2': writer lane (r0,n2,m0) reg r1 -> reader lane (r0,r1,m0) reg n2  [s_mid(T=r0,M=r1,L=m0,R=n2)]
    for l in range(64):
        T,R,L=dig(l)
        for Mm in range(4): xb[s_mid(T,Mm,L,R)]=regs2[l,Mm]
    regs3=np.zeros((64,4),complex)
    for l in range(64):
        T,Mm,L=dig(l)
        for R in range(4): regs3[l,R]=xb[s_mid(T,Mm,L,R)]
    # pass 3: lane (r0,r1,m0) reg n2 -> r2 ; twiddle W16^{m0 r2} (* W^{r} if p)
    for l in range(64):
        r0,r1,m0=dig(l)
        Y=d4(list(regs3[l]),False)
        for r2 in range(4):
            r=r0+4*r1+16*r2
            regs3[l,r2]=Y[r2]*W**(32*m0*r2)*(W**r if p else 1)
    # ex1': writer lane (r0,r1,m0) reg r2 -> reader lane (r0,r1,r2) reg m0 [s_low(T=r0,M=r1,L=r2,R=m0)]
    for l in range(64):
        T,Mm,R=dig(l)
        for L in range(4): xb[s_low(T,Mm,L,R)]=regs3[l,L]
    F=np.zeros(256,complex)
    for l in range(64):
        T,Mm,L=dig(l); r=T+4*Mm+16*L
        x=[xb[s_low(T,Mm,L,R)] for R in range(4)]
        Y=d4(x,False)
        for a in range(4): F[r+64*a]=Y[a]*(np.exp(2j*np.pi*a/8) if p else 1)
    return F

z=rng.normal(size=512)+1j*rng.normal(size=512)
A=np.array([sum(z[j]*psi**j*W**(j*k) for j in range(512)) for k in range(512)])
F0=forward_half(z,0); F1=forward_half(z,1)
A2=np.concatenate([F0+F1,F0-F1])
print('fwd err',np.abs(A-A2).max())
zz=np.array([psi**(-j)*sum(A[k]*W**(-j*k) for k in range(512)) for j in range(512)])
o0=inverse_half(A,0); o1=inverse_half(A,1)
got=np.zeros(512,complex)
for d in (o0,o1):
    for j,v in d.items(): got[j]=v
print('inv err',np.abs(got-zz).max(), np.abs(got/512-z).max())
# slot bijection & conflicts
WG=[list(range(g*8,g*8+8)) for g in range(8)]
RG=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
RG+= [[x+32 for x in g] for g in RG]
def conf(groups,slots,mod):
    w=1
    for g in groups:
        c={}
        for l in g: c.setdefault(slots[l]%mod,set()).add(slots[l])
        w=max(w,max(len(v) for v in c.values()))
    return w
def check(name,sf,wmap,rmap):
    # wmap(l,reg)->(T,M,L,R) for the writer; rmap likewise for reader
    ww=max(conf(WG,[sf(*wmap(l,reg)) for l in range(64)],8) for reg in range(4))
    rr=max(conf(RG,[sf(*rmap(l,reg)) for l in range(64)],16) for reg in range(4))
    print(name,'write-way',ww,'read-way',rr)
def W_low(l,R): T,Mm,L=dig(l); return (T,Mm,L,R)
def R_low(l,L): T,Mm,R=dig(l); return (T,Mm,L,R)
def W_mid(l,R): T,Mm,L=dig(l); return (T,Mm,L,R)
def R_mid(l,Mm): T,R,L=dig(l); return (T,Mm,L,R)
def W_top(l,R): T,Mm,L=dig(l); return (T,Mm,L,R)
def R_top(l,T): R,Mm,L=dig(l); return (T,Mm,L,R)
check('inv ex1 low',s_low,W_low,R_low); check('inv ex2 mid',s_mid,W_mid,R_mid); check('inv ex3 top',s_top,W_top,R_top)
check('fwd ex3 top',s_top,R_top,W_top); check('fwd ex2 mid',s_mid,R_mid,W_mid); check('fwd ex1 low',s_low,R_low,W_low)
