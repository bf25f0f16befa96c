"""Python host mirror of the reference's operator interfaces over the geglove C ABI.

Names and argument meaning follow the Java interfaces so parity tests read like reference
tests would (the reference has none):
  CoOccurrenceMatrix / BookmarkColoring   J/util/CoOccurrenceMatrix.java:6-17, J/bca/BookmarkColoring.java
  IOptimizer / Adagrad / Optimum           J/opt/IOptimizer.java:6-11, J/opt/grad/Adagrad.java, J/opt/Optimum.java
  GloveCost / PGloveCost                   J/opt/GloveCost.java, J/opt/PGloveCost.java
All compute goes through libgeglove.so (HIP, gfx950).  Nothing here falls back to the CPU.
"""
from . import capi
from .capi import GeError
from .host import (Configuration, CooMatrix, BookmarkColoring, Adagrad, Adam, AMSGrad, createOptimizer, Optimum,
                   GloveCost, PGloveCost, InvalidConfigurationException, SimilarityGroup, CompareGroup)

__all__ = ["capi", "GeError", "Configuration", "CooMatrix", "BookmarkColoring", "Adagrad", "Adam", "AMSGrad", "createOptimizer", "Optimum",
           "GloveCost", "PGloveCost", "InvalidConfigurationException", "SimilarityGroup", "CompareGroup"]
