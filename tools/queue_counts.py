#!/usr/bin/env python3
"""Candidates queued by the pyramid pass and keypoints accepted, per (octave, scale) group, for
one sparse and one dense synthetic 1080p image under both profiles (sift_get_group_counts)."""
import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd as sift
from bench import gen_images
for dense in (False, True):
    host = gen_images(1, 1000, dense, 4)
    for prof in (0,1):
        sift.set_params(n_octaves=5,n_scales=2,max_points=2500,profile=prof)
        d=torch.from_numpy(host[0]).cuda()
        pts=sift.new_point_buffer()
        sift.initDeviceVariables()
        sift.extractSift(pts,d,1920,1920,1080,None,0,0)
        n=sift.getPointCount(0)
        a,q=sift.group_counts(0)
        print('dense',dense,'profile',prof,'points',n,'accepted',a,'queued',q)
