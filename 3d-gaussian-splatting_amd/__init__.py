"""MI355X-native 3D Gaussian Splatting rasterizer (package ``gsplat_amd``)."""
