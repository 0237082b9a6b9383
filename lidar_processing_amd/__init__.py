"""lidar_processing_amd -- MI355X-native ground segmentation + obstacle clustering.

Host-side mirror of the reference's operator interface for the hot path
(`Segmenter`, `Clusterer`; reference src/segmentation.hpp:58-70, src/clustering.hpp:50-75) on top of
the C-ABI library `liblpx.so` (include/lpx.h).  There is no CPU fallback: without the HIP library
or without a GPU every compute call raises.
"""
from .api import (ClusteringConfiguration, Clusterer, Feeder, LpxError, PinnedArray, SegmentationConfiguration,
                  SegmentationLabel, Segmenter, Context, INVALID, UNDEFINED, load_pcd, pcd_info)
from .pcd import read_pcd, write_pcd

__all__ = ["ClusteringConfiguration", "Clusterer", "LpxError", "SegmentationConfiguration", "SegmentationLabel",
           "Segmenter", "Context", "INVALID", "UNDEFINED", "read_pcd", "write_pcd", "Feeder", "PinnedArray", "load_pcd",
           "pcd_info"]
