"""
Torch-free description of the reference U-Net's layer structure
(machine_learning/unet3d.py:53-75, 247-258 of the reference): which
convolutions exist, in state_dict order, with which channel counts.
"""


def unet_channels(width_multiplier=1):
    """
    Channel widths of the five U-Net levels (reference: unet3d.py:53-60).
    """
    return [int(c * width_multiplier) for c in (32, 64, 128, 256, 512)]


def unet_layer_specs(output_channels=1, trilinear=True, width_multiplier=1):
    """
    Lists the (prefix, in_channels, mid_channels, out_channels) of every
    DoubleConv in state_dict order, followed by the head's (in, out).

    Mirrors the constructor arithmetic of the reference's UNet3D/Down/Up
    (unet3d.py:53-75, 247-258); only "trilinear=True" is described.

    Returns
    -------
    Tuple[List[Tuple[str, int, int, int]], Tuple[int, int]]
    """
    if not trilinear:
        raise NotImplementedError("only trilinear=True is in scope")
    c = unet_channels(width_multiplier)
    f = 2
    blocks = [
        ("inc.double_conv", 1, c[0], c[0]),
        ("down1.maxpool_conv.1.double_conv", c[0], c[1], c[1]),
        ("down2.maxpool_conv.1.double_conv", c[1], c[2], c[2]),
        ("down3.maxpool_conv.1.double_conv", c[2], c[3], c[3]),
        ("down4.maxpool_conv.1.double_conv", c[3], c[4] // f, c[4] // f),
        ("up1.conv.double_conv", c[4], c[4] // 2, c[3] // f),
        ("up2.conv.double_conv", c[3], c[3] // 2, c[2] // f),
        ("up3.conv.double_conv", c[2], c[2] // 2, c[1] // f),
        ("up4.conv.double_conv", c[1], c[1] // 2, c[0]),
    ]
    return blocks, (c[0], output_channels)
