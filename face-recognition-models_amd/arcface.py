#!/usr/bin/env python3
"""Train ArcFace + ResNet-50 on CASIA-WebFace with the MI355X-native engine.
Same command line and outputs as the reference's main_code/arcface.py: stdout is duplicated into
WORKING_PATH/log/arcface.txt and checkpoints go to WORKING_PATH/checkpoints/ArcFace/."""
import os
import sys

from utils.config import DATASET_PATH, WORKING_PATH
from utils.criterion import ArcFaceNet
from utils.model_utils import main_pipeline
from utils.utils import Tee

if __name__ == "__main__":
    os.makedirs(f"{WORKING_PATH}/log", exist_ok=True)
    with open(f"{WORKING_PATH}/log/arcface.txt", "w") as log:
        sys.stdout = Tee(sys.__stdout__, log)
        main_pipeline(model_class=ArcFaceNet, model_name="ArcFace", project_name="face-recognition-training",
                      model_final_filename="arcface_final.pth", model_best_filename="arcface_best.pth",
                      num_classes=10575, working_path=WORKING_PATH, dataset_path=DATASET_PATH)
        sys.stdout = sys.__stdout__
