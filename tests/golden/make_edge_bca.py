import struct, sys, gzip, numpy as np
def read_bca(path):
    buf=open(path,'rb').read()
    magic,n,pos,lab=struct.unpack_from('<IQQQ',buf,0)
    lens=np.frombuffer(buf,np.uint32,n,pos)
    labels=buf[pos+4*n:pos+4*n+lab].split(b'\0')[:n]
    off=28; chains=[]
    for k in range(n):
        L=int(lens[k]); seq=buf[off:off+L]; ics=np.frombuffer(buf,np.uint16,3*L,off+L).reshape(L,3).copy(); off+=7*L
        chains.append((labels[k].decode(),seq,ics))
    return chains
def write_bca(path,chains):
    with open(path,'wb') as f:
        f.write(struct.pack('<IQQQ',0xBCABCA,len(chains),0,0))
        for lab,seq,ics in chains:
            f.write(seq); f.write(ics.astype(np.uint16).tobytes())
        pos=f.tell()
        f.write(np.array([len(c[1]) for c in chains],np.uint32).tobytes())
        labs=b''.join(c[0].encode()+b'\0' for c in chains)
        f.write(labs)
        f.seek(4); f.write(struct.pack('<QQQ',len(chains),pos,len(labs)))
q=read_bca('/root/reference/test_data/q100.bca'); p=read_bca('/root/reference/test_data/palms.bca')
out=[]
for k,L in enumerate([1,2,3,5,7,8,12,31,32,33,63,64,65]):
    lab,seq,ics=q[k]
    out.append(('tiny%d_%s'%(L,lab),seq[:L],ics[:L]))
out+=q[20:32]
plong=max(p,key=lambda c:len(c[1])); out.append(plong)
out.append(('dup_'+q[20][0],q[20][1],q[20][2]))           # identical chain under another label
out.append((q[21][0],q[21][1][::-1],q[21][2][::-1].copy()))  # same label as chain 21, reversed structure
write_bca(sys.argv[1],out)
print(len(out),[len(c[1]) for c in out])
