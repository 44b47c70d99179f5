"""Scene-grouped trajectory dataset (mirror of utils/dataloader.py:8-50; same output contract:
one item = (float32 [N, obs+pred, 2] in resized pixel coordinates, meta DataFrame, scene id))."""
import numpy as np
import torch
from torch.utils.data import Dataset


class SceneDataset(Dataset):
    def __init__(self, data, resize, total_len):
        self.trajectories, self.meta, self.scene_list = [], [], []
        for _, scene_df in data.groupby("sceneId", as_index=False):
            xy = scene_df[["x", "y"]].to_numpy().astype("float32").reshape(-1, total_len, 2)
            self.trajectories.append(xy * resize)
            self.meta.append(scene_df)
            self.scene_list.append(scene_df.iloc[0].sceneId)

    def __len__(self):
        return len(self.trajectories)

    def __getitem__(self, idx):
        return self.trajectories[idx], self.meta[idx], self.scene_list[idx]


def scene_collate(batch):
    trajectories = np.stack([item[0] for item in batch])
    return torch.from_numpy(trajectories).squeeze(0), [item[1] for item in batch], batch[0][2]
