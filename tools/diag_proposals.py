import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/x-detector_amd'); sys.path.insert(0,'/root/repo/tests')
from oracle import lighthead_oracle as O
from xdet import weights as W
from xdet.model import LightHeadDetector
from xdet.runtime import set_precision
w=W.make_lighthead_weights(1234)
imgs=W.synthetic_images(6,480,seed=20260928)
set_precision('f16x3')
det=LightHeadDetector(w,image_size=480,max_batch=6,rpn_post_nms_top_n=300,large_sep='direct')
set_precision('f32')
got=det.forward(imgs)
tr={}
ref=O.lighthead_forward(imgs,w,rpn_post_nms_top_n=300,trace=tr)
props=det.flat('proposals',(6,300,4))
obj=det.flat('objectness',(6,19800)); 
print('objectness max err', np.abs(obj-tr['objectness']).max(), 'rpn_boxes', np.abs(det.flat('rpn_boxes',(6,19800,4))-tr['rpn_boxes']).max())
for i in range(6):
    d=np.abs(props[i][:,None,:]-tr['proposals'][i][None,:,:]).max(-1)
    only_g=np.where(d.min(1)>=1e-3)[0]; only_o=np.where(d.min(0)>=1e-3)[0]
    print('image',i,'gpu-only',len(only_g),'oracle-only',len(only_o))
    for k in only_g:
        b=props[i][k]; both=np.concatenate([b[None],tr['proposals'][i]])
        ious=np.array([O.iou_tf(both,0,j) for j in range(1,len(both))])
        print('  gpu-only box',b,'closest |IoU-0.7|',np.abs(ious-0.7).min(), 'max iou', ious.max())
    for k in only_o:
        b=tr['proposals'][i][k]; both=np.concatenate([b[None],props[i]])
        ious=np.array([O.iou_tf(both,0,j) for j in range(1,len(both))])
        print('  oracle-only box',b,'closest |IoU-0.7|',np.abs(ious-0.7).min(), 'max iou', ious.max())
# ---- detections: which (image, class) lists differ, and how close to a threshold the difference sits
for i in range(6):
    for c in range(1,21):
        gs,gb=got[i][c]; rs,rb=ref[i][c]
        kg,kr=int((gs>0).sum()),int((rs>0).sum())
        used=np.zeros(kg,bool); un=[]
        for j in range(kr):
            d=np.where(used,np.inf,np.maximum(np.abs(gs[:kg]-rs[j]),np.abs(gb[:kg]-rb[j]).max(1))) if kg else np.array([np.inf])
            if d.min()<1e-3: used[int(d.argmin())]=True
            else: un.append(j)
        ex=[k for k in range(kg) if not used[k]]
        if un or ex:
            print('image',i,'class',c,'oracle-only',[(float(rs[j]),rb[j].tolist()) for j in un],'gpu-only',[(float(gs[k]),gb[k].tolist()) for k in ex])
            allb=np.concatenate([rb[:kr],gb[:kg]])
            for j in un:
                ious=np.array([O.iou_tf(np.stack([rb[j],b]),0,1) for b in gb[:kg]])
                print('   oracle-only det: IoUs with gpu dets closest to 0.3:', np.abs(ious-0.3).min(), 'max', ious.max())
            for k in ex:
                ious=np.array([O.iou_tf(np.stack([gb[k],b]),0,1) for b in rb[:kr]])
                print('   gpu-only det: IoUs with oracle dets closest to 0.3:', np.abs(ious-0.3).min(), 'max', ious.max(), 'score-0.01', gs[k]-0.01)
