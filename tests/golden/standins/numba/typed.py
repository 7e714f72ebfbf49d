List = list
