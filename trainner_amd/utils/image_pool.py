"""History buffer of generated images for the CycleGAN discriminators (codes/utils/image_pool.py:5-58).

Same draw sequence as the reference -- one `random.uniform(0, 1)` per image once the pool is full and one
`random.randint(0, pool_size - 1)` when a stored image is swapped out -- so a run seeded like the reference
(`random.seed`, utils/util.py set_random_seed) feeds its discriminators the same images.  The images stay in HBM.
"""
import random

import torch


class ImagePool:
    def __init__(self, pool_size):
        self.pool_size = int(pool_size or 0)
        self.num_imgs = 0
        self.images = []

    def query(self, images):
        """-> a batch the size of `images`: each entry is the new image or (with probability 1/2 once the pool is full)
        a stored one, which the new image then replaces.  The result carries no autograd history (`.data` in the reference)."""
        if self.pool_size == 0:
            return images
        out = torch.empty_like(images.detach())
        for i in range(images.shape[0]):
            image = images.detach()[i:i + 1]
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(image.clone())
                out[i:i + 1].copy_(image)
            elif random.uniform(0, 1) > 0.5:
                j = random.randint(0, self.pool_size - 1)
                out[i:i + 1].copy_(self.images[j])
                self.images[j] = image.clone()
            else:
                out[i:i + 1].copy_(image)
        return out
