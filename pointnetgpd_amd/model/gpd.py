"""Mirror of the reference's ``PointNetGPD/model/gpd.py`` — ``GPDClassifier``, the GPD baseline CNN the paper
compares against.  Same constructor, attribute names and construction order (identical state_dict
keys and seeded initialisation), same forward graph:

    conv1(C,20,5) -> pool1 -> conv2(20,50,5) -> pool2 -> view(-1,7200) -> relu(fc1) -> [dropout] -> fc2 -> log_softmax

Dispatch: CUDA tensors -> libpngpd, no fallback: eval mode runs ``pngpd_conv5_pool2`` x2 + ``pngpd_fc_fwd`` x2; train
mode runs the same stages as ONE autograd node (``gpd_ops.GPDNetFn``) whose backward — ``loss.backward()`` of
``main_1v_gpd.py:105`` — is ``pngpd_conv5_pool2_bwd`` x2, ``pngpd_fc_bwd`` x2, ``pngpd_relu_bwd`` and
``pngpd_log_softmax_bwd``.  CPU tensors -> the ATen composite the reference itself runs.  ``dropout=True`` in train mode
on CUDA raises (``nn.Dropout2d`` on a 2-D activation has no libpngpd kernel; the reference's default is off) unless the
ATen / MIOpen composite is opted into with ``GPDClassifier.allow_aten_training = True`` — never a silent dispatch."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import gpd_ops, ops


class GPDClassifier(nn.Module):
    """Input: (batch_size, input_chann, 60, 60)"""

    allow_aten_training = False      # opt-in: CUDA + train() + dropout=True runs the ATen / MIOpen composite

    def __init__(self, input_chann, dropout=False):
        super().__init__()
        self.conv1 = nn.Conv2d(input_chann, 20, 5)
        self.pool1 = nn.MaxPool2d(2, stride=2)
        self.conv2 = nn.Conv2d(20, 50, 5)
        self.pool2 = nn.MaxPool2d(2, stride=2)
        self.fc1 = nn.Linear(12 * 12 * 50, 500)
        self.dp = nn.Dropout2d(p=0.5, inplace=False)
        self.relu = nn.ReLU()
        self.fc2 = nn.Linear(500, 2)
        self.if_dropout = dropout

    def forward(self, x):
        if x.is_cuda and not self.training:
            return self._forward_hip(x)
        if x.is_cuda and not (self.if_dropout and self.allow_aten_training):
            if self.if_dropout:
                raise RuntimeError("GPDClassifier: dropout=True in train() mode has no libpngpd kernel (nn.Dropout2d on "
                                   "the (B,500) activation, gpd.py:28-29); a CUDA tensor would run on ATen/MIOpen. Opt "
                                   "in with GPDClassifier.allow_aten_training = True, construct with dropout=False, "
                                   "or train on the CPU.")
            self._check_shape(x)
            return gpd_ops.GPDNetFn.apply(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                          self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        x = self.pool1(self.conv1(x))
        x = self.pool2(self.conv2(x))
        x = x.view(-1, 7200)
        x = self.relu(self.fc1(x))
        if self.if_dropout:
            x = self.dp(x)
        x = self.fc2(x)
        return F.log_softmax(x, dim=-1)

    def _check_shape(self, x):
        if x.dim() != 4 or x.shape[1] != self.conv1.in_channels or x.shape[2] != 60 or x.shape[3] != 60:
            raise RuntimeError(f"expected input of shape (B,{self.conv1.in_channels},60,60), got {tuple(x.shape)}")

    def _forward_hip(self, x):
        self._check_shape(x)
        x = x.float().contiguous()
        x = gpd_ops.conv5_pool2(x, self.conv1.weight, self.conv1.bias)          # (B,20,28,28)
        x = gpd_ops.conv5_pool2(x, self.conv2.weight, self.conv2.bias)          # (B,50,12,12)
        x = x.view(-1, 7200)
        x = gpd_ops.fc_fwd_splitk(x, self.fc1.weight.detach().contiguous(), self.fc1.bias.detach().contiguous(), True)
        return ops.fc_fwd(x, self.fc2.weight.detach().contiguous(), self.fc2.bias.detach().contiguous(),
                          ops.EPI_LOG_SOFTMAX)                                   # K = 500 = 8 * 62 + 4: the kernel's tail
