import sys; sys.path.insert(0,'.')
import torch, numpy as np
import recnet_amd as R
from tests.test_search_oracle import load_search_case
from oracle import recnet_oracle as O
g,P,enc=load_search_case("search_gru_stop")
B,F,D,V,E,H,A=[int(x) for x in g["meta_dims"]]
dec=R.Decoder("GRU",1,D,E,1,H,A,V,0.5,0.5,0.5,precision="f32"); dec.load_state_dict(P); dec=dec.cuda().eval()
tok=torch.full((1,B),1,dtype=torch.long)
with torch.no_grad():
    lg0,h0=O.decoder_step(P,tok,O.zero_hidden(B,H,"GRU"),enc,cell="GRU",t=0)
    dl0,dh0=dec(tok.cuda(),torch.zeros(1,B,H,device="cuda"),enc.cuda())
    print("step0 logits diff",(dl0.cpu()-lg0).abs().max().item(),"h diff",(dh0.cpu()-h0).abs().max().item())
    for t2 in (0,49,28):
        tk=torch.full((1,B),t2,dtype=torch.long)
        lg1,h1=O.decoder_step(P,tk,h0,enc,cell="GRU",t=1)
        dl1,dh1=dec(tk.cuda(),dh0,enc.cuda())
        print(t2,"step1 logits diff",(dl1.cpu()-lg1).abs().max().item(),"h diff",(dh1.cpu()-h1).abs().max().item(), lg1[:,0])
    class Cfg: caption_max_len=30; decoder_model="GRU"; batch_size=B
    for bw in (1,2,3):
        print(bw, R.beam_search(Cfg(),bw,None,dec,tok.cuda(),torch.zeros(1,B,H,device="cuda"),enc.cuda()))
