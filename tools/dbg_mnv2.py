import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import face_recognition_amd as fr
from oracle import backbones as ob
net = ob.seeded('mobilenetv2', 0).eval()
m = fr.get_network('mobilenetv2'); m.load_state_dict(net.state_dict()); m = m.cuda().eval()
names = m.layer_names()
i = names.index('features.17.conv.3')
for seed in (0, 1, 2):
    torch.manual_seed(seed)
    x = (torch.rand((8,3,112,112))*2-1)
    _, taps = ob.forward_with_taps(net, x, ['features.17.conv.3', 'features.17.conv.1'])
    ref = taps['features.17.conv.3']
    outs = []
    for rep in range(3):
        _, tap = m.forward_tap(x.cuda(), i, 'bf16')
        outs.append(tap.cpu())
    cos = torch.nn.functional.cosine_similarity(outs[0].flatten(), ref.flatten(), dim=0)
    print('seed', seed, 'cos', float(cos), 'rep diffs', float((outs[0]-outs[1]).abs().max()), float((outs[0]-outs[2]).abs().max()), 'ref max', float(ref.abs().max()))
    d = (outs[0]-ref).abs()
    print('  err by image', [float(v) for v in d.amax(dim=(1,2,3))])
