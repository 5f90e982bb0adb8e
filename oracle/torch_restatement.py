"""TEST / BASELINE INFRASTRUCTURE -- torch-functional restatement of the reference forward.

The reference's CPU path *is* torch ATen (oneDNN convolutions): `DeepSpeakerModel.forward`
(/root/reference/model.py:185-218) is a chain of nn.Conv2d / nn.BatchNorm2d / nn.Hardtanh /
nn.AdaptiveAvgPool2d / nn.Linear calls.  This file restates that chain with torch.nn.functional
so that bench.py can time "the reference's own CPU forward" on the GPU box's host cores, where
/root/reference does not exist (cpu_baseline.kind = "port").  It is pinned against the recorded
reference outputs by tests/test_oracle_golden.py::test_torch_restatement.  Never imported by the
product package.
"""
import torch
import torch.nn.functional as F


def forward_eval(sd, x, n_stages: int = 4):
    """sd: reference-keyed state_dict of torch tensors; x: [B,1,T,64] float32 CPU tensor."""
    def bn(t, name):
        return F.batch_norm(t, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], False, 0.1, 1e-5)                     # model.py:188 (eval)

    for i in range(1, n_stages + 1):
        x = F.conv2d(x, sd[f"model.conv{i}.weight"], None, 2, 2)                      # model.py:187,192,197,202
        x = F.hardtanh(bn(x, f"model.bn{i}"), 0.0, 20.0)                              # model.py:188-189
        r = x                                                                         # model.py:67
        y = F.conv2d(x, sd[f"model.layer{i}.0.conv1.weight"], None, 1, 1)             # model.py:69
        y = F.hardtanh(bn(y, f"model.layer{i}.0.bn1"), 0.0, 20.0)                     # model.py:70-71
        y = F.conv2d(y, sd[f"model.layer{i}.0.conv2.weight"], None, 1, 1)             # model.py:73
        y = bn(y, f"model.layer{i}.0.bn2")                                            # model.py:74
        x = F.hardtanh(y + r, 0.0, 20.0)                                              # model.py:79-80
    x = F.adaptive_avg_pool2d(x, (1, None))                                           # model.py:207
    x = x.view(x.size(0), -1)                                                         # model.py:208
    x = F.linear(x, sd["model.fc.weight"], sd["model.fc.bias"])                       # model.py:209
    norm = torch.sqrt(torch.sum(x * x, 1) + 1e-10)                                    # model.py:174-177
    return x / norm.view(-1, 1) * 10                                                  # model.py:179,212
