"""``nn`` namespace of the reference (``eval(f"nn.{name}")(*params)``, reference
src/mnist_exm.py:424; ``from nn import ...`` at :24-25).  The reference ships no
``nn/__init__.py`` (only a 1-byte ``nn/init``), so the export list is defined here."""
from .qconv import QConv2d
from .qdense import (QDenseUndirected_old, QDenseUndirected_old_noise, QIDDM_L, QIDDM_LL_noise,
                     QIDDM_LL_relu_noise, QIDDM_PL, QIDDM_PL_noise, QNN, QNN_A, QNN_noise,
                     differN_noise, differN_noise_befor, differN_old_pca)
from .qdense_more import (QIDDM_A_differN_NEW, QIDDM_A_differN_basePL, QIDDM_A_sameN, QIDDM_CL_new,
                          QIDDM_CL_old, QIDDM_L_B, QIDDM_LL_old, QIDDM_PL_noise1, QIDDM_PL_old, QIDDM_PP_noise,
                          QIDDM_PP_old, QIDDM_bias_false, differN_new_conv, differN_new_pca, differN_old_conv)
from .unet import Conv2d, DownBlock, UNetUndirected, UnetDirected, UpBlock
from .unet_simple import DownBlockS, UNetUndirectedS, UnetDirectedS, UpBlockS
from .utils import autocrop, autopad, get_label_embedding

__all__ = [
    "QConv2d", "QDenseUndirected_old", "QDenseUndirected_old_noise", "QIDDM_L", "QIDDM_LL_noise",
    "QIDDM_LL_relu_noise", "QIDDM_PL", "QIDDM_PL_noise", "QNN", "QNN_A", "QNN_noise", "differN_noise",
    "differN_noise_befor", "differN_old_pca", "QIDDM_A_differN_NEW", "QIDDM_A_differN_basePL", "QIDDM_A_sameN",
    "QIDDM_CL_new", "QIDDM_CL_old", "QIDDM_L_B", "QIDDM_LL_old", "QIDDM_PL_noise1", "QIDDM_PL_old",
    "QIDDM_PP_noise", "QIDDM_PP_old", "QIDDM_bias_false", "differN_new_conv", "differN_new_pca",
    "differN_old_conv", "Conv2d", "DownBlock", "UNetUndirected", "UnetDirected",
    "UpBlock", "DownBlockS", "UNetUndirectedS", "UnetDirectedS", "UpBlockS", "autocrop", "autopad",
    "get_label_embedding",
]
