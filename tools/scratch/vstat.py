import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/x-detector_amd')
from xdet import weights as W
from xdet.model import LightHeadDetector
w=W.make_lighthead_weights(1234)
imgs=W.synthetic_images(4,480,seed=1)
for R in (300,1000):
    det=LightHeadDetector(w,image_size=480,max_batch=4,rpn_post_nms_top_n=R)
    got=det.forward(imgs)
    cr=det.buffer('cls_reg',4).numpy().reshape(4,R,-1)
    lg=cr[...,:21]; e=np.exp(lg-lg.max(-1,keepdims=True)); p=e/e.sum(-1,keepdims=True)
    V=(p[...,1:]>0.01).sum(1)
    print('R',R,'V per (image,class): mean %.1f max %d'%(V.mean(),V.max()), 'per image keep', det.flat('prop_counts',(4,4),np.int32)[:,2] if False else '')
    nd=[[int((got[i][c][0]>0).sum()) for c in range(1,21)] for i in range(4)]
    print(' dets per class mean', np.mean(nd), 'max', np.max(nd))
