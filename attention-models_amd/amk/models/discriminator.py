"""PatchGAN discriminator used by the ViT-VQGAN train step (reference:
models/utils/discriminator.py:6-54, constructed as NLayerDiscriminator(3, 64, 3) in
trainers/vitgqgan.py:64).  Not a kernel target: convolutions / BatchNorm stay on MIOpen.
It exists here because the benchmark step contains it; state_dict
keys (``model.N.*``) match the reference layer order."""
import torch.nn as nn


class NLayerDiscriminator(nn.Module):
    def __init__(self, input_nc=3, ndf=64, n_layers=3):
        super().__init__()
        widths = [ndf * min(2 ** i, 8) for i in range(n_layers + 1)]   # 64, 128, 256, 512
        layers = [nn.Conv2d(input_nc, widths[0], 4, stride=2, padding=1), nn.LeakyReLU(0.2, True)]
        for i in range(1, n_layers + 1):
            stride = 2 if i < n_layers else 1
            layers += [
                nn.Conv2d(widths[i - 1], widths[i], 4, stride=stride, padding=1, bias=False),
                nn.BatchNorm2d(widths[i]),
                nn.LeakyReLU(0.2, True),
            ]
        layers.append(nn.Conv2d(widths[-1], 1, 4, stride=1, padding=1))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)
