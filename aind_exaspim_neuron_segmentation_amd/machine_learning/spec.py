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
    Lists the parameterised layers of the network in state_dict order.

    Mirrors the constructor arithmetic of the reference's UNet3D/Down/Up
    (unet3d.py:53-75, 247-258). Entries are

    * ("double_conv", prefix, in_channels, mid_channels, out_channels) for every
      DoubleConv, and
    * ("conv_transpose", prefix, in_channels, out_channels) for the
      ConvTranspose3d(k=2, s=2) of each Up block when "trilinear" is False
      (registered before the block's DoubleConv, unet3d.py:254-258).

    Returns
    -------
    Tuple[List[tuple], Tuple[int, int]]
        The layer list and the head's (in_channels, out_channels).
    """
    c = unet_channels(width_multiplier)
    f = 2 if trilinear else 1
    layers = [
        ("double_conv", "inc.double_conv", 1, c[0], c[0]),
        ("double_conv", "down1.maxpool_conv.1.double_conv", c[0], c[1], c[1]),
        ("double_conv", "down2.maxpool_conv.1.double_conv", c[1], c[2], c[2]),
        ("double_conv", "down3.maxpool_conv.1.double_conv", c[2], c[3], c[3]),
        ("double_conv", "down4.maxpool_conv.1.double_conv", c[3], c[4] // f, c[4] // f),
    ]
    ups = [("up1", c[4], c[3] // f), ("up2", c[3], c[2] // f), ("up3", c[2], c[1] // f),
           ("up4", c[1], c[0])]
    for name, cin, cout in ups:
        if trilinear:
            layers.append(("double_conv", f"{name}.conv.double_conv", cin, cin // 2, cout))
        else:
            layers.append(("conv_transpose", f"{name}.up", cin, cin // 2))
            layers.append(("double_conv", f"{name}.conv.double_conv", cin, cout, cout))
    return layers, (c[0], output_channels)
