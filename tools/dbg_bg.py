import sys; sys.path.insert(0, '.')
import torch
from egopack_amd import data as D, ops
from egopack_amd.graphone import build_graphone
from egopack_amd.models import Graph
from egopack_amd.models.tasks import LTATask, PNRTask, RecognitionTask
from oracle import path as O, pyg_ops as P
G = torch.load('tests/golden/build_graphone.pt', weights_only=False)
trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": 40}
m = Graph(48, hidden_size=32, depth=3, temporal_pooling=trn, num_segments=3); m.load_state_dict(G['backbone']); m.cuda().eval()
ar = RecognitionTask(32, 32, G['n_classes']); ar.load_state_dict(G['tasks']['ar']); ar.cuda().eval()
ops.set_compute('f32')
with torch.no_grad():
    for bi, b in enumerate(G['batches']):
        d = D.Data(**b).to('cuda')
        feat = m(d)
        ref = O.graph_forward(G['backbone'], b['x'], b['pos'], b['edge_index'], 3)
        print(bi, 'feat diff', (feat.cpu()-ref).abs().max().item())
        tf = ar.forward_features(feat)
        rtf = O.projection_features(G['tasks']['ar'], ref)
        print(bi, 'taskfeat diff', (tf.cpu()-rtf).abs().max().item())
        y = d.y
        labels = torch.where(y[:, 0] != -1, y[:, 0] * 11 + y[:, 1], torch.full_like(y[:, 0], -1))
        print(labels[labels>=0].tolist())
        bank = torch.zeros(77, 32, dtype=torch.float64, device='cuda'); cnt = torch.zeros(77, dtype=torch.int64, device='cuda')
        ops.scatter_add_rows_f64(tf, labels, bank, cnt)
        keep = (b['y'][:,0] != -1)
        rb = P.scatter_sum(rtf[keep].double(), (b['y'][keep][:,0]*11+b['y'][keep][:,1]), 77)
        print(bi, 'bank diff', (bank.cpu()-rb).abs().max().item(), cnt.sum().item())
banks = build_graphone(m, ar, [ar], [D.Data(**b) for b in G['batches']], device='cuda')
print((banks['ar'].cpu()-G['banks']['ar']).abs().max())
