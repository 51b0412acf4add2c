from .base import TestDataset, TestDatasetRelation, TrainDataset
from .dataset import Dataset
from .device import DeviceBatches
from .named import CountriesS1, Fb15k237, Umls, Wn18rr, Yago310

__all__ = ["CountriesS1", "Dataset", "DeviceBatches", "Fb15k237", "TestDataset", "TestDatasetRelation", "TrainDataset", "Umls",
           "Wn18rr", "Yago310"]
